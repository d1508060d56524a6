#!/bin/bash
# Runs ON THE GPU BOX (gpurun): kernel-trace summaries and PMC passes of bench.py for the round-2 build.
# Everything lands under gpurun_out/r03/; tools/make_counters_json.py turns it into profiles/r03_*.
export TMPDIR=/tmp
out=gpurun_out/r03
mkdir -p $out
common="--steps 50 --warmup 100 --no-cpu-baseline --no-host-rate --no-reference-leg --bank-workers 1 --lanes 1"
trace() {  # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py $common "$@" > $out/trace_$name.json 2> $out/trace_$name.err || { tail -3 $out/trace_$name.err; return 1; }
  f=$(find $out/trace_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/r03_${name}_kernel_stats.csv
  echo "trace $name done"
}
pmc() {  # name, pass, counters, bench args
  name=$1; pass=$2; ctrs=$3; shift 3
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_${name}_$pass -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-rate --no-reference-leg --bank-workers 1 --lanes 1 "$@" > $out/pmc_${name}_$pass.log 2>&1 || { tail -3 $out/pmc_${name}_$pass.log; return 1; }
  python3 - "$out/pmc_${name}_$pass" > $out/r03_pmc_${name}_$pass.txt <<'PY'
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        m = re.search(r"dysf_[a-z0-9_]+_kernel(<[^>]*>)?", kn)
        if m:
            k = m.group(0)
            a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k}\t{c}\t{v / n:.1f}\t{n}")
PY
  rm -rf $out/pmc_${name}_$pass
  echo "pmc $name $pass done"
}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU"
SQ3="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES"
if [ "$1" = "cnn-only" ]; then pmc cnn sq1 "$SQ1" --head cnn && pmc cnn sq2 "$SQ2" --head cnn; ls $out; exit 0; fi
trace bench &&
trace reference_149 --preset reference_149 &&
trace cnn --head cnn &&
pmc bench sq1 "$SQ1" && pmc bench sq2 "$SQ2" && pmc bench sq3 "$SQ3" && pmc bench fetch "FETCH_SIZE" && pmc bench write "WRITE_SIZE" &&
pmc reference_149 sq1 "$SQ1" --preset reference_149 && pmc reference_149 sq2 "$SQ2" --preset reference_149 &&
pmc reference_149 fetch "FETCH_SIZE" --preset reference_149 && pmc reference_149 write "WRITE_SIZE" --preset reference_149 &&
pmc cnn sq1 "$SQ1" --head cnn && pmc cnn sq2 "$SQ2" --head cnn
rm -rf $out/trace_bench $out/trace_reference_149 $out/trace_cnn
ls $out
