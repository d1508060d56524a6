#!/usr/bin/env python3
"""Turns gpurun_out/rNN/ (written on the GPU box by tools/collect_rNN.sh) into the committed profile artefacts:
profiles/rNN_*_kernel_stats.csv, profiles/rNN_pmc_*.txt (per-kernel per-launch counter averages) and
profiles/rNN_counters.json (derived figures that bench.py quotes, stamped with the id of the kernel sources).
  python tools/make_counters_json.py [r03]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # kernel_source_id()
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC, DST = os.path.join(ROOT, "gpurun_out", TAG), os.path.join(ROOT, "profiles")
N_SIMD, N_CU, N_XCD = 1024, 256, 8


def read_pmc(name):
    out = {}
    for f in glob.glob(os.path.join(SRC, f"{TAG}_pmc_{name}_*.txt")):
        shutil.copy(f, DST)
        for line in open(f):
            k, c, v, n = line.rstrip("\n").split("\t")
            out.setdefault(k.split("<")[0], {})[c] = float(v)
    return out


def durations(name):
    f = os.path.join(SRC, f"{TAG}_{name}_kernel_stats.csv")
    shutil.copy(f, DST)
    d = {}
    for r in csv.DictReader(open(f)):
        if "dysf" in r["Name"]:
            import re
            k = re.search(r"dysf_[a-z0-9_]+_kernel", r["Name"]).group(0)
            d[k] = d.get(k, 0.0) + float(r["AverageNs"]) * 1e-3   # us (distinct instantiations of one name add up)
    return d


def derive(c, us):
    """Per-kernel derived figures.  SQ_* wave counters are in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and
    SQ_LDS_IDX_ACTIVE are cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, rocprofv3 section)."""
    r = {"avg_us_kernel_trace": round(us, 1)}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        r["wave_time_split"] = {"issuing": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), "issue_stalled": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                                "parked_at_waitcnt_or_barrier": round(c["SQ_WAIT_ANY"] / wc, 3)}
    gui = c.get("GRBM_GUI_ACTIVE")
    cycles = gui / N_XCD if gui else us * 1e-6 * 2.2e9   # kernel length in shader cycles
    r["shader_cycles"] = round(cycles)
    if gui and us > 0:
        r["effective_clock_ghz"] = round(cycles / (us * 1e-6) / 1e9, 2)
    if "SQ_ACTIVE_INST_VALU" in c:
        r["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * N_SIMD), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        r["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * N_SIMD), 3)
    if "SQ_LDS_IDX_ACTIVE" in c:
        r["lds_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / (cycles * N_CU), 3)
        r["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 3)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_WAVES"):
        if k in c:
            r[k.lower()] = int(c[k])
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced (16 B per lane)
        # streaming reads -- all bulk reads of these kernels are such loads -- so it is doubled; WRITE_SIZE as is
        r["hbm_read_bytes"] = int(c.get("FETCH_SIZE", 0.0) * 1024 * 2)
        r["hbm_write_bytes"] = int(c.get("WRITE_SIZE", 0.0) * 1024)
    return r


def limited(r):
    s = r.get("wave_time_split", {})
    return ("not HBM: the vector pipe -- VALU %.0f %% busy (a wave instruction holds its SIMD's 16-lane pipe for 4 cycles; 4.1-4.6 measured, "
            "tools/ubench/valu_rate.hip), LDS %.0f %%, MFMA %.0f %%; waves spend %.0f %% of their time parked at s_waitcnt and %.0f %% "
            "issue-stalled; the floor is the vector-instruction count"
            % (100 * r.get("valu_busy", 0), 100 * r.get("lds_busy", 0), 100 * r.get("mfma_busy", 0),
               100 * s.get("parked_at_waitcnt_or_barrier", 0), 100 * s.get("issue_stalled", 0)))


out = {"round": int(TAG[1:]), "kernel_source_id": bench.kernel_source_id(),
       "how": "tools/collect_" + TAG + ".sh on one MI355X: rocprofv3 --kernel-trace --stats, and separate rocprofv3 --kernel-trace --pmc passes "
              "(SQ counters in groups of 8, FETCH_SIZE and WRITE_SIZE each alone) of `python bench.py --lanes 1 --no-cpu-baseline "
              "--no-host-rate --bank-workers 1 [--preset reference_149 | --head cnn]`; per-launch averages in profiles/" + TAG + "_pmc_*.txt"}
for name in ("bench", "reference_149", "cnn"):
    pm, du = read_pmc(name), durations(name)
    kern = {}
    for k, c in pm.items():
        if not k.startswith("dysf_"):
            continue
        kern[k] = derive(c, du.get(k, 0.0))
    feat = [k for k in kern if k.startswith(("dysf_stft", "dysf_mfcc", "dysf_chroma"))]
    dom = next((k for k in kern if k.startswith("dysf_stft")), None) if name != "cnn" else next((k for k in kern if "conv2" in k), None)
    entry = {"kernel_source_id": out["kernel_source_id"], "source": f"profiles/{TAG}_pmc_{name}_*.txt, profiles/{TAG}_{name}_kernel_stats.csv",
             "kernels": kern}
    if all("hbm_read_bytes" in kern[k] for k in feat) and feat:
        entry["hbm_bytes_per_launch"] = sum(kern[k]["hbm_read_bytes"] + kern[k]["hbm_write_bytes"] for k in feat)
        entry["hbm_bytes_per_launch_is"] = "PMC traffic of the feature launch (" + " + ".join(feat) + "), reads corrected x2"
    if dom and dom in kern:
        entry["dominant_kernel"] = dom
        entry["valu_busy"], entry["mfma_busy"] = kern[dom].get("valu_busy"), kern[dom].get("mfma_busy")
        entry["limited_by"] = limited(kern[dom])
    out["bench_512" if name == "bench" else name] = entry
json.dump(out, open(os.path.join(DST, TAG + "_counters.json"), "w"), indent=1)
for name in ("bench", "reference_149", "cnn"):
    f = os.path.join(SRC, f"trace_{name}.json")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(DST, f"{TAG}_{name}_bench_under_rocprof.json"))
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} if isinstance(v, dict) else v for k, v in out.items()}, indent=1))
print(json.dumps(out["bench_512"]["kernels"], indent=1))
