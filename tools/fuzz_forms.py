"""robustness sweep: random extents / light counts / ring depths / shardings / root relief / frames per launch / wave-plan targets; the fused
frame (after enough frames for the wave plan to have switched) must equal the per-ray staged frame bit for bit"""
import os, sys, random
sys.path.insert(0, ".")
import numpy as np
from araytracingjourney_amd import renderer as R, scenes
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cor, spo = scenes.cornell(), scenes.sponza_like(0.05)
lights16 = scenes.sponza_lights(4) * 4
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for case in range(n_cases):
    sc = random.choice((cor, spo))
    w, h = random.choice((1, 7, 31, 32, 33, 64, 100, 257, 320)), random.choice((1, 8, 9, 32, 47, 96, 130, 200))
    nl = random.choice((0, 1, 2, 4, 7, 16))
    fif = random.choice((1, 2, 5))
    G = random.choice((1, 1, 2, 3))
    k = random.randrange(G)
    packed = G > 1 and random.random() < 0.5
    B = random.choice((1, 1, 2, 3, 4))
    relief = random.choice((0, 0, 40, 200)) if G > 1 else 0
    beam = random.choice(({}, {}, {"packet_wide": 2}))   # the packets walk the 4-wide nodes (default) or the binary ones
    split = random.choice((8, 30, 100000))   # wave-plan target in packet steps (ArtTuning.split_fixed_steps): nearly everything / some / nothing splits
    outs = []
    for form in ("fused", "per-ray"):
        r = R.renderer_for_scene(sc, (w, h), n_lights=0, shard=(k, G), frames_in_flight=fif, packed_tiles=packed, fast_build=random.random() < 0.3, root_relief=relief,
                                 tuning={"frame_form": 2} if form == "per-ray" else dict(beam, split_fixed_steps=split))
        if form == "fused" and B > 1: r.set_frames_per_launch(B)
        for d in (lights16[:nl] if sc is spo else [dict(sc.lights[0], pos=(0.1 * i - 0.3, 0.5, 0.05 * i)) for i in range(nl)]):
            r.lights_mut().push_dict(d)
        for i in range(fif + 1 if form == "per-ray" else 2 * fif + 3):
            r.render_frame()
            if form == "fused": r.sync()
        if form == "fused" and B > 1: r.set_read_frame(random.randrange(B))
        st = r.stats()
        outs.append((r.read_color(), r.read_depth(), r.read_normal(), r.read_color_tiles() if G > 1 else None, st["shadow_rays"], st["hit_pixels"], st["primary_rays"], st["split_blocks"]))
        r.close()
    a, b = outs
    ok = all(np.array_equal(a[i].view(np.uint32), b[i].view(np.uint32)) for i in range(3)) and a[4:7] == b[4:7] and (G == 1 or np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)))
    print(f"case {case}: {sc.name} {w}x{h} lights {nl} F {fif} shard {k}/{G} relief {relief} packed {packed} B {B} split {split} nodes {beam} (blocks split: {a[7]}): {'ok' if ok else 'MISMATCH'} rays {a[6]}+{a[4]}", flush=True)
    if not ok: sys.exit(1)
print("FUZZ_OK")
