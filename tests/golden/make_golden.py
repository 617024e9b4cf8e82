"""Regenerates the committed fixtures from the CPU oracle (oracle/liborc.so).  Run from the repo root:
    python tests/golden/make_golden.py [--full]
The reference itself cannot run here (Rust + Vulkan RT), so these are the oracle's own outputs:
they pin the oracle against regressions and give bench.py the canonical-LBVH visit counters of SURVEY.md 8(d).
--full also recomputes the 1080p / 4K stats of BASELINE configs 2 and 3 (a few CPU-seconds each)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from araytracingjourney_amd import scenes  # noqa: E402
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def render(sc, w, h, lights, threads=8, bits=30):
    S = orc.Scene(sc.primitives, morton_bits=bits)
    cam = orc.camera_from_params(sc.camera["pos"], sc.camera["dir"], w / h, sc.camera["fovy"], sc.camera["znear"], sc.camera["zfar"])
    return S.render(cam, orc.make_lights(lights), len(lights), w, h, threads=threads)


def main():
    c = scenes.cornell()
    for n in (64, 256):
        out = render(c, n, n, c.lights)
        np.savez_compressed(os.path.join(HERE, f"cornell_{n}.npz"), color=out["color"], depth=out["depth"], normal=out["normal"])
        json.dump(out["stats"], open(os.path.join(HERE, f"cornell_{n}.stats.json"), "w"), indent=1)
    if "--full" in sys.argv:
        s = scenes.sponza_like()
        for tag, (w, h), lights in (("c2_sponza_like_1080p_1light", (1920, 1080), scenes.sponza_lights(1)),
                                    ("c3_sponza_like_2160p_4lights", (3840, 2160), scenes.sponza_lights(4))):
            out = render(s, w, h, lights)
            st = dict(out["stats"], width=w, height=h, n_lights=len(lights), n_tris=s.n_tris, morton_bits=30)
            json.dump(st, open(os.path.join(HERE, f"{tag}.stats.json"), "w"), indent=1)
            print(tag, st)


if __name__ == "__main__":
    main()
