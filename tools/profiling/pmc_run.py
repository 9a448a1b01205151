"""Workload of the round-2 counter passes (tools/profiling/pmc_r2.sh): the headline scene with ONE kernel per batched substep
(R2S_CHAINS=1: a dispatch = the whole 32-env batch), two env steps of free motion and three in contact, each followed by the
render; preceded by a 512 MiB copy with known HBM byte counts (calibration of FETCH_SIZE / WRITE_SIZE)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import torch

from r2s_hip.rollout import BatchedRollout


def main():
    x = torch.empty(512 * 1024 * 1024 // 4, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)          # reads 512 MiB, writes 512 MiB per call
    torch.cuda.synchronize()
    rate = float(os.environ.get("PMC_CLOSE_RATE", "0"))   # round 6: 0.1 = the closing ramp (the grasp latches around env step 11: PMC_STEPS=14 ends in the held grasp)
    ro = BatchedRollout(os.environ.get("PMC_CONFIG", "sloth_32env"), close_at=2, settle_steps=int(os.environ.get("PMC_SETTLE", "12")), close_rate=rate if rate > 0 else None)
    for _ in range(int(os.environ.get("PMC_STEPS", "5"))):
        ro.step()
    torch.cuda.synchronize()
    print("contact stats at the end:", ro.contact_stats())


if __name__ == "__main__":
    main()
