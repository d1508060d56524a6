#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (SQ counters in groups of <= 8) of tools/kdrive.py; per-kernel per-launch averages.
# usage: tools/pmc_k.sh <tag> [kdrive args]      output: gpurun_out/pmc_<tag>_{sq1,sq2,sq3}.txt
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU"
SQ3="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
for pass in sq1 sq2 sq3; do
  case $pass in sq1) c="$SQ1";; sq2) c="$SQ2";; sq3) c="$SQ3";; esac
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmcdir_${tag}_$pass -- python3 tools/kdrive.py 6 "$@" > $out/pmc_${tag}_$pass.log 2>&1 || { tail -3 $out/pmc_${tag}_$pass.log; exit 1; }
  python3 - "$out/pmcdir_${tag}_$pass" > $out/pmc_${tag}_$pass.txt <<'PY'
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"dysf_[a-z0-9_]+_kernel(<[^>]*>)?", r["Kernel_Name"])
        if m:
            a = agg[(m.group(0), r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k}\t{c}\t{v / n:.1f}\t{n}")
PY
  rm -rf $out/pmcdir_${tag}_$pass
done
cat $out/pmc_${tag}_sq*.txt
