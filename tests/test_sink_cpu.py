"""Host half of the observation sink that needs no GPU: the episode-video step (experiments/utils/ffmpeg.py:5-21)."""
import os
import stat


def test_make_video_issues_the_reference_ffmpeg_command_line(tmp_path, monkeypatch):
    """ffmpeg is an external program for the reference too (eval_policy.py:261-267 shells out to it).  With a stand-in
    executable on PATH the worker's make_video must issue exactly the reference's arguments; without one it reports False."""
    from r2s_hip import _sink_worker

    rgb = tmp_path / "episode_0000" / "camera_0" / "rgb"
    rgb.mkdir(parents=True)
    video = tmp_path / "episode_0000" / "vis_camera_0.mp4"
    monkeypatch.setenv("PATH", str(tmp_path / "nobin"))
    assert _sink_worker.make_video(rgb, video, "%06d.jpg", 10) is False
    bindir = tmp_path / "bin"
    bindir.mkdir()
    log = tmp_path / "args.txt"
    fake = bindir / "ffmpeg"
    fake.write_text(f"#!/bin/sh\nfor a in \"$@\"; do echo \"$a\" >> {log}; done\n: > \"$(eval echo \\${{$#}})\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{bindir}:/usr/bin:/bin")
    assert _sink_worker.make_video(rgb, video, "%06d.jpg", 10) is True
    args = log.read_text().split("\n")[:-1]
    assert args == ["-y", "-hide_banner", "-loglevel", "error", "-framerate", "10", "-i", os.path.join(str(rgb), "%06d.jpg"), "-c:v", "libx264", "-pix_fmt", "yuv420p",
                    str(video)]
    assert video.exists()
