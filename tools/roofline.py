#!/usr/bin/env python3
"""Re-derives every number of bench.py's `roofline` object from what is committed under profiles/.

    python tools/roofline.py --tag round3 [--set-current]

Inputs (all under profiles/, all produced on the GPU box by tools/profile_round.sh: tools/pmc.sh passes and plain bench runs), per BASELINE config that was profiled:
    config 2   <tag>_pmc.txt      <tag>_bench_line.json   (+ <tag>_kernel_stats_bench_c2.csv: rocprofv3 --kernel-trace --stats of `bench.py --steps 1000`)
    config 3   <tag>_pmc_c3.txt   <tag>_config3.json
    config 4   <tag>_pmc_c4.txt   <tag>_config4.json
    config 5   <tag>_pmc_c5.txt   <tag>_config5.json      (dominant kernel: the AO launch's k_trace_ao -- k_trace<4, 4> until round 3)
(a pmc file: one line per kernel and pass, name {counter: mean} n=launches) and tests/golden/<config>.stats.json (the oracle's visit counters of that frame).

What it prints (and writes to profiles/<tag>_roofline.json; --set-current also writes profiles/current_pmc.json, the file bench.py reads
its instruction counts and HBM traffic from -- PMC counters cannot be read from inside the benchmarked process):

    valu_issue   SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz) / time per launch      <- the binding roof (roofline.frac)
    salu_issue   SQ_INSTS_SALU x 1 cycle  / ( 256 CUs   x 2.4 GHz) / time per launch
    hbm          (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B / time per launch / 8 TB/s         (MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 64 B per 128-B request)
    packet       oracle packet-level algorithmic bytes / time per launch / 8 TB/s          (a node / triangle once per 8x8-pixel packet)
    contract     SURVEY.md 8(d) per-ray algorithmic bytes / time per launch / 8 TB/s       (> 1: every ray is charged for nodes its packet fetches once)

"time per launch" is the machine's time per frame at steady state = ms_per_step of the bench line (sixteen launches overlap, so one launch's
own span -- kernel_ms, the stats CSV's average -- is ~13 x longer and is NOT what the chip spends on it)."""
import argparse
import ast
import csv
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_HZ = 2.4e9          # MI355X peak engine clock (MI355X_MICROARCH.md)
SIMDS, CUS = 1024, 256
VALU_CYCLES = 2           # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (transcendentals, f32 divides' v_rcp / v_sqrt: 4)
HBM_PEAK = 8.0e12


def kernel_source_hash():
    """what the PMC counts belong to: the frame kernels' sources"""
    h = hashlib.sha256()
    for f in ("art_trace.hip", "art_internal.h"):
        h.update(open(os.path.join(ROOT, "araytracingjourney_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(st, n_lights):
    """per frame, from the oracle's counters on the canonical LBVH (tests/golden/*.stats.json)"""
    shade = (24 + 12 + 144 + 48 + 80 * n_lights) * st["hit_pixels"]           # PrimitiveInfo + indices + 3 vertices + 3 bilinear quads + lights, per hit pixel
    contract = (32 + 16) * st["primary_rays"] + 64 * st["n_int_primary"] + 48 * st["n_tri_primary"] + (32 + 4) * st["shadow_rays"] + 64 * st["n_int_shadow"] + 48 * st["n_tri_shadow"] \
        + shade + 24 * st["primary_rays"]
    out = dict(contract=contract)
    if "packet_nodes_primary" in st:
        nodes, tris = st["packet_nodes_primary"] + st["packet_nodes_shadow"], st["packet_tris_primary"] + st["packet_tris_shadow"]
        # a packet tracer: every node (64 B) and triangle (48 B) once per 8x8 packet; rays, hits and contributions stay in registers; the frame's
        # outputs as the API defines them (RGBA32F colour + F32 depth + RGBA32F normal = 36 B per pixel)
        out.update(packet=64 * nodes + 48 * tris + shade + 36 * st["primary_rays"], packet_nodes=nodes, packet_tris=tris, packet_traversal_bytes=64 * nodes + 48 * tris,
                   shading_bytes=shade, output_bytes=36 * st["primary_rays"])
    return out


def fractions(pmc, us_per_launch, ab):
    """pmc: per-launch means (SQ_INSTS_VALU, SQ_INSTS_SALU, FETCH_SIZE, WRITE_SIZE in KB, ...); us_per_launch: machine time per launch"""
    t = us_per_launch * 1e-6
    r = {}
    if "SQ_INSTS_VALU" in pmc:
        r["valu_issue_us"] = pmc["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * CLOCK_HZ) * 1e6
        r["valu_issue_frac"] = r["valu_issue_us"] / us_per_launch
    if "SQ_INSTS_SALU" in pmc:
        r["salu_issue_us"] = pmc["SQ_INSTS_SALU"] / (CUS * CLOCK_HZ) * 1e6
        r["salu_issue_frac"] = r["salu_issue_us"] / us_per_launch
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc and us_per_launch > 0:
        r["hbm_bytes_per_launch"] = int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
        r["hbm_frac"] = r["hbm_bytes_per_launch"] / t / HBM_PEAK
    if "TCC_HIT_sum" in pmc and pmc.get("TCC_REQ_sum"):
        r["l2_hit_rate"] = pmc["TCC_HIT_sum"] / pmc["TCC_REQ_sum"]
    if "SQ_THREAD_CYCLES_VALU" in pmc and pmc.get("SQ_ACTIVE_INST_VALU"):
        r["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64)
    if ab:
        r["contract_frac"] = ab["contract"] / t / HBM_PEAK
        if "packet" in ab:
            r["packet_frac"] = ab["packet"] / t / HBM_PEAK
    return r


def parse_pmc_txt(path, kernel_substr="k_frame"):
    """tools/pmc.sh output: `<kernel name tail> {counter: mean, ...} n=<launches>` per kernel and pass; the instance with the most launches wins"""
    best = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s*(\{.*\})\s*n=(\d+)\s*$", line)
        if not m:
            continue
        name, counters, n = m.group(1), ast.literal_eval(m.group(2)), int(m.group(3))
        if kernel_substr == "k_frame":
            if kernel_substr not in name and not re.search(r"\d, (true|false), (true|false)", name):   # the name is cut to its tail: k_frame<...> shows as its template arguments
                continue
            if "k_frame_stats" in name:
                continue
        elif kernel_substr not in name:
            continue
        for c, v in counters.items():
            if c not in best or n > best[c][1]:
                best[c] = (v, n)
    return {c: v for c, (v, n) in best.items()}, max((n for _, n in best.values()), default=0)


def parse_stats_csv(path, kernel_substr="k_frame<"):
    rows = [r for r in csv.DictReader(open(path)) if kernel_substr in r["Name"]]
    if not rows:
        return None
    top = max(rows, key=lambda r: int(r["Calls"]))
    return dict(name=top["Name"], calls=int(top["Calls"]), average_ns=float(top["AverageNs"]), min_ns=float(top["MinNs"]), max_ns=float(top["MaxNs"]))


CONFIGS = {  # what a profile round covers: name -> (file suffixes under profiles/<tag>_*, dominant kernel as tools/pmc.sh prints it, golden stats, lights)
    "c2": dict(pmc="pmc.txt", line="bench_line.json", kernel="k_frame", golden="c2_sponza_like_1080p_1light", lights=1),
    "c3": dict(pmc="pmc_c3.txt", line="config3.json", kernel="k_frame", golden="c3_sponza_like_2160p_4lights", lights=4),
    "c4": dict(pmc="pmc_c4.txt", line="config4.json", kernel="k_frame", golden="c4_bistro_like_1080p_1light", lights=1),
    "c5": dict(pmc="pmc_c5.txt", line="config5.json", kernel=("k_trace_ao", "k_trace<4, 4>"), golden="c5_sponza_like_2160p_16spp_ao", lights=1),   # round 4: the AO launch's own tracer; before: the generic one
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--set-current", action="store_true", help="also write profiles/current_pmc.json (what bench.py reads: one entry per workload that has a counter pass)")
    a = ap.parse_args()
    P = os.path.join(ROOT, "profiles")
    out_all, workloads = {}, {}
    for name, c in CONFIGS.items():
        fp, fl = os.path.join(P, f"{a.tag}_{c['pmc']}"), os.path.join(P, f"{a.tag}_{c['line']}")
        if not (os.path.exists(fp) and os.path.exists(fl)):
            continue
        kernels = c["kernel"] if isinstance(c["kernel"], tuple) else (c["kernel"],)
        for kname in kernels:
            pmc, launches = parse_pmc_txt(fp, kname)
            if launches:
                break
        c = dict(c, kernel=kname)
        line = json.load(open(fl))
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", c["golden"] + ".stats.json")))
        if name == "c5":   # the AO launch: its machine time is the step minus the frames without their AO pass (bench.py measures both)
            rl = line.get("roofline") or {}
            us = rl.get("machine_us_per_launch") or line["ms_per_step"] * 1e3
            ab = dict(contract=(32 + 1) * gold["ao_rays"] + 64 * gold["n_int_ao"] + 48 * gold["n_tri_ao"])
        else:
            us = line["ms_per_step"] * 1e3
            ab = algorithmic_bytes(gold, c["lights"])
        fr = fractions(pmc, us, ab)
        fta = os.path.join(P, f"{a.tag}_pmc_ta_{name}.txt")   # tools/pmc_ta.sh: the vector-memory path's own counters
        if os.path.exists(fta):
            ta, n_ta = parse_pmc_txt(fta, kname)
            if n_ta and ta.get("GRBM_GUI_ACTIVE"):
                # TA_TA_BUSY_sum: over the 256 addressers (one a CU); GRBM_GUI_ACTIVE: over the 8 XCDs -- both of the kernel's own launches under the profiler
                fr["ta_busy_frac"] = (ta["TA_TA_BUSY_sum"] / CUS) / (ta["GRBM_GUI_ACTIVE"] / 8)
                if ta.get("TA_FLAT_READ_WAVEFRONTS_sum"):
                    fr["ta_cycles_per_load_instruction"] = ta["TA_TA_BUSY_sum"] / ta["TA_FLAT_READ_WAVEFRONTS_sum"]
                    if "TCP_TOTAL_CACHE_ACCESSES_sum" in ta:
                        fr["l1_accesses_per_load_instruction"] = ta["TCP_TOTAL_CACHE_ACCESSES_sum"] / ta["TA_FLAT_READ_WAVEFRONTS_sum"]
                if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in ta:
                    fr["ta_stalled_by_l1_frac"] = (ta["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / CUS) / (ta["GRBM_GUI_ACTIVE"] / 8)
                pmc = dict(pmc, **ta)
        cands = [k for k in ("valu_issue_frac", "salu_issue_frac", "hbm_frac", "ta_busy_frac") if k in fr]
        bound = max(cands, key=lambda k: fr[k]).replace("_frac", "") if cands else None
        out = dict(tag=a.tag, config=name, workload=line["config"]["workload"], kernel=c["kernel"], us_per_launch_machine=us, ms_per_step=line["ms_per_step"], mray_per_s=line["value"],
                   pmc_launches_averaged=launches, pmc=pmc, algorithmic_bytes_per_frame=ab, binding_roof=bound, **fr)
        if "valu_issue_frac" in fr and "valu_lane_utilisation" in fr:
            out["useful_valu_frac"] = fr["valu_issue_frac"] * fr["valu_lane_utilisation"]
        if name == "c2":
            sp = os.path.join(P, f"{a.tag}_kernel_stats_bench_c2.csv")
            stats = parse_stats_csv(sp) if os.path.exists(sp) else None
            if stats:
                out.update(kernel=stats["name"], kernel_ms_rocprof_average=stats["average_ns"] * 1e-6, kernel_ms_bench_events=(line.get("roofline") or {}).get("kernel_ms"),
                           launches_overlapping=stats["average_ns"] * 1e-3 / us)
        out_all[name] = out
        workloads[line["config"]["workload"]] = dict(tag=a.tag, config=name, kernel=c["kernel"], kernel_source_sha16=kernel_source_hash(), pmc=pmc,
                                                     source=f"profiles/{a.tag}_{c['pmc']} (rocprofv3 --kernel-trace --pmc, separate passes, tools/pmc.sh; means over {launches} launches of {c['kernel']})")
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("pmc",)} for k, v in out_all.items()}, indent=1))
    json.dump(out_all, open(os.path.join(P, f"{a.tag}_roofline.json"), "w"), indent=1)
    if a.set_current:
        json.dump(dict(tag=a.tag, workloads=workloads), open(os.path.join(P, "current_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
