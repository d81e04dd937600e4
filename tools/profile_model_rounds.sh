#!/bin/bash
# Per-kernel times of the model rounds (1-11) of rsem-run-em on a generated PairedEndQModel input, for the per-read model
# kernel (MODES="default lib:<tag>": variant builds beside it; the older kernel families left the library in round 6).   tools/profile_model_rounds.sh [n_reads] [M]
N=${1:-5263157}; M=${2:-200000}; D=/tmp/pmr
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $D; tools/bin/gen_temp $D $N $M 3 20250925 100 nosam 5-16 | tail -1
tools/bin/temp_to_rsb $D/temp/s $D/stat/s 3 > /dev/null
# a mode "lib:<tag>" runs the default kernels of the variant build rsem_amd/librsem_hip_<tag>.so (built like tools/build_variants.sh builds its own, with model.hip's object replaced)
for mode in ${MODES:-default}; do
  export RSEM_HIP_NORMAL_EXIT=1
  unset LD_LIBRARY_PATH
  if [ $mode = default ]; then unset RSEM_MODEL_KERNELS
  elif [[ $mode == lib:* ]]; then
    unset RSEM_MODEL_KERNELS; tag=${mode#lib:}; mkdir -p /tmp/pmr_lib_$tag; cp rsem_amd/librsem_hip_$tag.so /tmp/pmr_lib_$tag/librsem_hip.so
    export LD_LIBRARY_PATH=/tmp/pmr_lib_$tag; mode=lib_$tag
  else export RSEM_MODEL_KERNELS=$mode; fi
  rm -rf gpurun_out/pmr_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmr_$mode -o p -- rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -q > /dev/null 2>&1
  echo "== RSEM_MODEL_KERNELS=$mode"
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/pmr_$mode/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:10]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-44s calls %5s avg %10.1f us total %8.1f ms" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf $D
