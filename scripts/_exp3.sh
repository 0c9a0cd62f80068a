cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $OUT
for lib in base p_base; do
for wh in "3840 2160" "3904 2160" "3856 2160" "3776 2160" "4096 2160"; do
  export OAVIF_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ablate/$lib/liboavif_hip.so
  set -- $wh
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rg_bench.py $1 $2 > $OUT/b.log 2>&1 || { echo fail; tail -3 $OUT/b.log; exit 1; }
  f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
  echo "== $lib $wh: $(grep 'rg_bench:' $OUT/b.log | cut -c1-90)"
  python3 - $f $1 $2 <<'PY'
import csv,sys
w,h=int(sys.argv[2]),int(sys.argv[3])
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "k_rg_" in n or "bands_xyb" in n:
        avg=float(r["AverageNs"])/1e3
        print(f"   {n[:40]:40s} {avg:8.1f} us   {avg/(w*h/1e6):6.2f} us/MP")
PY
  rm -rf $OUT/p
done; done
