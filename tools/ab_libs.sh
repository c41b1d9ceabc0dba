# usage (ON THE GPU BOX): bash tools/ab_libs.sh "<workload list>" name1=lib1.so name2=lib2.so ...   ("-" = the in-tree library)
# Same-session A/B of builds of libexcenv_hip.so: every workload runs under every library, twice, interleaved.
set -e
WL="$1"; shift
mkdir -p gpurun_out/ab
for w in $WL; do
  for rep in 1 2; do
    for kv in "$@"; do
      name=${kv%%=*}; lib=${kv#*=}
      if [ "$lib" = "-" ]; then
        python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab/${w}_${name}_$rep.json 2>> gpurun_out/ab/err.txt
      else
        EXCENV_HIP_LIB=$PWD/$lib python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab/${w}_${name}_$rep.json 2>> gpurun_out/ab/err.txt
      fi
    done
  done
  for kv in "$@"; do
    name=${kv%%=*}
    python - <<PY
import json
r=[json.load(open("gpurun_out/ab/${w}_${name}_%d.json"%i)) for i in (1,2)]
print("%-22s %-10s"%("$w","$name"), "  ".join("%.3f ms frac %.3f"%(x["ms_per_step"],x["roofline"]["frac"]) for x in r))
PY
  done
done
