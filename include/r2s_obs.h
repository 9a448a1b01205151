/*
 * r2s_obs.h — C ABI of the device half of the observation sink.
 *
 * The reference converts every rendered camera image on the host, one image at a time, inside the episode loop:
 *   image = (image.cpu().numpy().transpose(1, 2, 0) * 255).astype(np.uint8); image = cv2.cvtColor(image, cv2.COLOR_RGB2BGR)
 *   experiments/eval_policy.py:157-158 (and :249-250), after torch.clamp(im, 0, 1) in GSRenderer.render (gs_renderer.py:949)
 * i.e. a blocking 3.7 MB float D2H per 640x480 frame plus numpy passes.  Here ONE kernel packs all frames of a batch
 * (environments x cameras) into interleaved 8-bit pixels — 4x fewer bytes to move — and the Python host (r2s_hip/sink.py)
 * copies them to pinned ring buffers asynchronously and hands them to a writer thread (JPEG encoding, state pickles).
 */
#ifndef R2S_OBS_H
#define R2S_OBS_H

#include <stddef.h>
#include <stdint.h>
#include "r2s_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

/* color: device float32 [n_frames, 3, H, W] planar RGB as the rasteriser writes it (out_color of every frame, contiguous);
 * out: device uint8 [n_frames, H, W, 3] interleaved, channel order BGR when bgr != 0 (what cv2.imwrite takes) else RGB.
 * Each value = (uint8)(min(max(c, 0), 1) * 255.0f) — clamp (gs_renderer.py:949), float32 product, truncation (astype). */
int r2s_obs_pack_u8(const float* color, int32_t n_frames, int32_t height, int32_t width, int32_t bgr, uint8_t* out,
                    r2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* R2S_OBS_H */
