#!/usr/bin/env python3
"""The reference's interactive step, end to end: every frame it recomputes the camera (Layer::update_camera /
Raytracer::set_render_params, layer.rs:188-193, mod.rs:353-388) and adds `num_samples_per_pixel` = 2 samples per pixel
(mod.rs:605-613).  Here: mirt_ctx_set_camera (host-side only: the camera travels by value with the launch) followed by
a 2-spp render into device memory, queued back to back on one stream, 1920x1080, config-3 scene.

    python tools/interactive_step.py [frames]      -> prints one JSON object (also written to gpurun_out/)

Both timing modes of the context (mirt_ctx_set_timing): ON = every launch carries an event pair (kernel times in the statistics),
OFF = the frames carry no event at all (what a host that only displays the frames wants).  And DOUBLE-BUFFERED: frames alternate between
the context's two frame streams (mirt_ctx_frame_stream: different hardware queues) and two framebuffers, as a swap chain's frames do --
the head of frame k + 1 then fills the GPU while the tail of frame k drains.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch  # noqa: E402
import weekend_raytracer_wgpu_amd as m  # noqa: E402
from helpers import scene_data  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
w, h = 1920, 1080
sd = scene_data("three_spheres", w, h)
ctx = m.Context(0)
ctx.set_scene(sd)
fc = m.FlyCameraController.default()
cams = []
for i in range(8):                                        # a small orbit: a different camera every frame
    c = fc.renderer_camera()
    c.eye_pos = (np.asarray(c.eye_pos, np.float32) + np.float32(0.01 * i) * np.array([1, 0, 0.5], np.float32)).astype(np.float32)
    cams.append(m.GpuCamera.new(c, (w, h)).c)
outs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [ctx.frame_stream(0), ctx.frame_stream(1)]          # the context's two frame streams: different hardware queues by construction
res = {}
for timing, n_buf in ((True, 1), (False, 1), (True, 2), (False, 2)):
  ctx.set_timing(timing)
  name = ("timing on (default)" if timing else "timing off (mirt_ctx_set_timing(ctx, 0))") + (", double-buffered (mirt_ctx_frame_stream 0 / 1, two framebuffers)" if n_buf == 2 else "")
  mode = res.setdefault(name, {})
  for label, spp in (("2 spp per frame (reference default)", 2), ("1 spp per frame", 1), ("4 spp per frame", 4)):
    p = m.make_params(w, h, spp, mode=m.MIRT_MODE_PT, num_bounces=8)
    for i in range(20):                                   # warm-up
        ctx.set_camera(cams[i % 8])
        ctx.render_device(p, outs[i % n_buf].data_ptr(), outs[0].numel(), streams[i % n_buf])
    torch.cuda.synchronize()
    ctx.stats()
    t0 = time.perf_counter()
    for i in range(frames):
        ctx.set_camera(cams[i % 8])
        ctx.render_device(p, outs[i % n_buf].data_ptr(), outs[0].numel(), streams[i % n_buf])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    st = ctx.stats()
    mode[label] = {"frames": frames, "host_issue_us_per_frame": round(1e6 * t_issue / frames, 2),
                   "end_to_end_us_per_frame": round(1e6 * t_all / frames, 2),
                   "kernel_us_per_frame_hip_events": round(1e3 * st["kernel_ms_total"] / max(1, st["launches"]), 2) if timing else None,
                   "msamples_per_s_end_to_end": round(w * h * spp * frames / t_all / 1e6, 1), "kernel": ctx.last_kernel()}
print(json.dumps(res, indent=1))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "interactive_step.json").write_text(json.dumps(res, indent=1) + "\n")
ctx.close()
