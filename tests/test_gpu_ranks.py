"""bench.py's N > 1 path on the one GPU of a test box: two ranks share the card and gather through
host memory (BENCH_BACKEND=gloo); RCCL itself needs one GPU per rank.  What is checked is what a
multi-GPU node runs apart from the transport: device selection per rank, frames of the sequence dealt
out to the ranks, interleaved row groups of one 16-spp frame per rank, the gathers, the max-over-ranks
timing and the one JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_print_one_line_with_both_scalings(gpu):
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    common = ["--steps", "4", "--warmup", "2", "--repeats", "0", "--frames16", "4", "--width", "320", "--height", "184", "--mesh", "teapot",
              "--texture-size", "256", "--no-cpu-baseline", "--no-serial-leg", "--no-configs", "--checksums"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    # the same frames on one rank: both partitions must deliver the same presented frames, byte for byte
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sequence-leg", "--only", "f16w"] + common,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert one.returncode == 0, one.stderr[-2000:]
    alone = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])["frame_16spp"]["whitted_area_light"]
    two = out["frame_16spp"]["whitted_area_light"]
    for leg in ("pipelined", "sequence"):
        a, b = alone[leg]["checksums"], two[leg]["checksums"]
        shared = sorted(set(a) & set(b))
        # (the orbit camera is inside the teapot, which is much larger than the backpack: most frames are dark)
        assert len(shared) >= 2 and all(a[k] == b[k] for k in shared) and any(a[k] > 0 for k in shared), (leg, a, b)
    assert two["sequence"]["partition"] == "frames" and two["sequence"]["frames"] == 8 and two["sequence"]["n_gpus"] == 2
    assert out["strong_scaling"]["whitted_area_light"]["sequence_ms_per_frame"] == two["sequence"]["ms_per_frame"]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["steps"] == 4 and out["warmup"] == 2
    for kind in ("whitted_area_light", "path_ggx"):
        f = out["frame_16spp"][kind]
        assert f["n_gpus"] == 2 and f["scaling"] == "strong" and f["ms_per_frame"] > 0
        assert out["strong_scaling"][kind]["n_gpus"] == 2
        assert f["pipelined"]["n_gpus"] == 2 and f["pipelined"]["frames_in_flight"] == 3 and f["pipelined"]["ms_per_frame"] > 0
        assert out["strong_scaling"][kind]["pipelined_ms_per_frame"] == f["pipelined"]["ms_per_frame"]
    # the rays of the frames the two ranks rendered, summed: more than one rank's share of the steps
    assert out["rays_per_frame"] > 320 * 184


@pytest.mark.gpu
def test_one_rank_through_rccl(gpu):
    """The N > 1 code path of bench.py -- process group with a device id, asynchronous gathers of the
    presented frames on the frames' streams, barriers, reductions -- with RCCL itself as the transport and
    a single rank (BENCH_DIST=1), which is all a one-GPU box can offer RCCL."""
    env = dict(os.environ, BENCH_DIST="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    env.pop("BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--repeats", "0", "--frames16", "4", "--width", "320", "--height", "184", "--mesh", "teapot", "--texture-size", "256",
           "--no-cpu-baseline", "--no-serial-leg", "--no-configs", "--sequence-leg", "--checksums"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    for kind in ("whitted_area_light", "path_ggx"):
        f = out["frame_16spp"][kind]
        assert f["ms_per_frame"] > 0 and f["pipelined"]["ms_per_frame"] > 0 and f["sequence"]["ms_per_frame"] > 0
