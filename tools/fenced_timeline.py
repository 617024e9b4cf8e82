#!/usr/bin/env python3
"""When does each frame of a short fenced run (the driver's `bench.py --steps 20 --warmup 5`) finish?

The GPU is idle at the first fence; K frames are launched back to back; the host then polls art_frames_done per frame and notes the time
each one is first seen finished.  Prints the launch time of every frame (host, since the fence) and its completion time, so that the fill
of the ring, the steady part and the drain can be told apart.

    python tools/fenced_timeline.py [--steps 20] [--frames-in-flight 16] [--tuning hw_queues=4,...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from araytracingjourney_amd import renderer, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-in-flight", type=int, default=16)
    ap.add_argument("--tuning", default="")
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--offset", type=int, default=0, help="extra frames once, before the first run (moves the ring slot every run starts in)")
    ap.add_argument("--quiet", action="store_true", help="one summary line")
    ap.add_argument("--no-poll", action="store_true", help="no per-frame polling: art_sync only (what bench.py's fence does)")
    a = ap.parse_args()
    sc = scenes.sponza_like()
    sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(1))
    tuning = {k: (float(v) if k == "split_alpha" else int(v)) for k, v in (kv.split("=") for kv in a.tuning.split(",") if kv)} or None
    F = a.frames_in_flight
    r = renderer.renderer_for_scene(sc, (1920, 1080), frames_in_flight=F, tuning=tuning)
    r.upload_state()
    for _ in range(3 * F):
        r.trace()
    for _ in range(a.offset):
        r.trace()
    r.sync()
    rays = None
    walls = []
    for rep in range(a.repeat):
        for _ in range(a.warmup):
            r.trace()
        r.sync()
        base = r.frames_traced()
        t0 = time.perf_counter()
        launched, done = [], [None] * a.steps
        for i in range(a.steps):
            r.trace()
            launched.append(time.perf_counter() - t0)
        nxt = a.steps if a.no_poll else 0
        if a.no_poll:
            done = [0.0] * a.steps
        while nxt < a.steps:          # frames may finish out of order: scan from the first unfinished one
            for i in range(nxt, a.steps):
                if done[i] is None and r.frames_done(base + i, 1):
                    done[i] = time.perf_counter() - t0
            while nxt < a.steps and done[nxt] is not None:
                nxt += 1
        r.sync()
        wall = time.perf_counter() - t0
        if rays is None:
            st = r.stats()
            rays = st["primary_rays"] + st["shadow_rays"]
        walls.append(wall)
        if a.quiet:
            continue
        print(f"run {rep}: {a.steps} frames fenced in {wall * 1e3:.3f} ms = {rays * a.steps / wall / 1e6:,.0f} Mray/s   (last launch returned at {launched[-1] * 1e3:.3f} ms)")
        if rep == a.repeat - 1 and not a.quiet:
            for i in range(a.steps):
                print(f"   frame {i:2d}: launched {launched[i] * 1e3:7.3f} ms   finished {done[i] * 1e3:7.3f} ms")
    if a.quiet:
        walls.sort()
        print(f"first slot of the last run {(base) % F:2d}: median {walls[len(walls) // 2] * 1e3:.3f} ms = {rays * a.steps / walls[len(walls) // 2] / 1e6:,.0f} Mray/s, best {walls[0] * 1e3:.3f}, worst {walls[-1] * 1e3:.3f}")


if __name__ == "__main__":
    main()
