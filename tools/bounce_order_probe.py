#!/usr/bin/env python3
"""What the ORDER of incoherent rays is worth to the BVH walk (run on the MI355X box).

Camera rays of the backpack scene (1920x1080, 2 per pixel) are traced, a random direction of the
hemisphere over the hit's normal makes one bounce ray per hit (seeded numpy: a probe, not the
renderer's random stream), and the same set of about two million bounce rays goes through
rtx_trace_closest with the render path's walk (k_trace_api<3>) in different orders:

  queue         the order of the parents' pixels
  bins          64 octahedral direction bins, queue order kept inside a bin
  bins+cells    direction bins, inside a bin by an 8x8x8 grid cell of the origin
  random        a random permutation
  by-steps      the random order sorted by the number of steps each ray needs (every lane of a wave
                busy to the end, no coherence at all)

Per order it prints lanes busy = sum of the rays' steps / sum over waves of 64 x the longest ray's
steps, from the per-ray work counters.  Kernel times come from a kernel trace of the same run:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/bounce_probe -- python3 $REPO/tools/bounce_order_probe.py
    python3 $REPO/tools/bounce_order_probe.py --times $REPO/gpurun_out/bounce_probe

(the k_trace_api<3> launches after the first are the orders, in the order listed above).
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ORDERS = ["queue", "bins", "bins+cells", "random", "by-steps"]


def camera_rays(cam, W, H, per_pixel, rng):
    eye = np.array(list(cam.eye), np.float64)
    right, up, fwd = (np.array(list(v), np.float64) for v in (cam.right, cam.up, cam.forward))
    ys, xs = np.mgrid[0:H, 0:W]
    xs = np.repeat(xs.reshape(-1), per_pixel) + rng.uniform(0, 1, W * H * per_pixel)
    ys = np.repeat(ys.reshape(-1), per_pixel) + rng.uniform(0, 1, W * H * per_pixel)
    d = (xs / W - 0.5)[:, None] * right + (ys / H - 0.5)[:, None] * up + fwd * cam.focal
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.broadcast_to(eye, d.shape).copy(), d


def octahedral_bin(d, n=8):
    a = d / np.abs(d).sum(axis=1)[:, None]
    u, v = a[:, 0].copy(), a[:, 1].copy()
    neg = a[:, 2] < 0
    u[neg] = (1 - np.abs(a[neg, 1])) * np.sign(a[neg, 0] + 1e-300)
    v[neg] = (1 - np.abs(a[neg, 0])) * np.sign(a[neg, 1] + 1e-300)
    iu = np.clip(((u * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1)
    iv = np.clip(((v * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1)
    return iu * n + iv


def lanes_busy(steps):
    pad = (-len(steps)) % 64
    s = np.concatenate([steps, np.zeros(pad, steps.dtype)]).reshape(-1, 64)
    return float(s.sum() / (64.0 * s.max(axis=1).sum()))


def times(directory):
    rows = []
    for f in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_trace_api<3>" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    rows.sort()
    for name, (_, ms) in zip(["camera rays"] + ORDERS, rows):
        print(json.dumps({"launch": name, "kernel_ms": round(ms, 3)}))


def main(argv):
    if "--times" in argv:
        return times(argv[argv.index("--times") + 1])
    from raytracer_amd import abi, device, scenes
    device.init(0)
    rng = np.random.default_rng(2024)
    W, H = 1920, 1080
    desc = scenes.frame16_scene("path").frame(0.0)
    rd = scenes.frame16_render_desc("path", W, H, 1)
    o, d = camera_rays(rd.camera, W, H, 2, rng)
    with device.DeviceScene(desc, static=True) as ds:
        first = ds.trace_closest(o, d, flags=abi.FLAG_RENDER_WALK)
        hit = first["prim"] >= 0
        nrm = first["normal"][hit]
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        out = rng.normal(size=nrm.shape)
        out /= np.linalg.norm(out, axis=1)[:, None]
        out[(out * nrm).sum(axis=1) < 0] *= -1.0
        bo = first["position"][hit] + nrm * 1e-3
        n = len(bo)
        print(json.dumps({"camera_rays": len(o), "bounce_rays": n}), flush=True)
        dbin = octahedral_bin(out)
        lo, hi = bo.min(axis=0), bo.max(axis=0)
        cell = np.clip(((bo - lo) / np.maximum(hi - lo, 1e-12) * 8).astype(np.int64), 0, 7)
        cell = (cell[:, 0] * 8 + cell[:, 1]) * 8 + cell[:, 2]
        shuffled = rng.permutation(n)
        perms = {"queue": np.arange(n), "bins": np.argsort(dbin, kind="stable"),
                 "bins+cells": np.lexsort((cell, dbin)), "random": shuffled}
        steps_of = None
        for name in ORDERS:
            if name == "by-steps":
                perm = shuffled[np.argsort(steps_of[shuffled], kind="stable")]
            else:
                perm = perms[name]
            got = ds.trace_closest(bo[perm], out[perm], flags=abi.FLAG_RENDER_WALK, want_counters=True)
            c = got["counters"].astype(np.int64)
            # per ray: node visits (the first counter holds two child-box tests each, plus the few KD
            # branches the certificate looks at) + shape tests
            steps = c[:, 0] // 2 + c[:, 3] + c[:, 4]
            if steps_of is None:
                steps_of = np.empty(n, np.int64)
            steps_of[perm] = steps
            q = np.quantile(steps, [0.5, 0.99])
            print(json.dumps({"order": name, "lanes_busy": round(lanes_busy(steps), 3), "steps_mean": round(float(steps.mean()), 1),
                              "steps_median": float(q[0]), "steps_p99": float(q[1]), "steps_max": int(steps.max()),
                              "hits": int((got["prim"] >= 0).sum())}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
