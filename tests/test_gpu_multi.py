"""N > 1 on the GPU box (one GPU): two real ranks through the pipelined TiledFrame (tests/two_ranks_one_gpu.py)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,in_flight", [(2, 1), (4, 1), (2, 2), (4, 2)])
def test_real_ranks_through_the_pipelined_frame_on_one_gpu(world, in_flight):
    """Two / four processes, the HIP kernels, double-buffered parts, an asynchronous gather per frame (gloo on CUDA tensors: RCCL refuses two
    ranks on one device), de-interleave on the device: seven frames with different seeds, each equal to the single-rank render.  in_flight = 2:
    consecutive frames render on the context's two frame streams (what bench.py's N > 1 runs do)."""
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "two_ranks_one_gpu.py"), "7", str(world), str(in_flight)], capture_output=True, text=True,
                       timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert f"{world} ranks ok: 7 pipelined frames" in r.stdout and f"({in_flight} in flight)" in r.stdout, r.stdout[-2000:]
