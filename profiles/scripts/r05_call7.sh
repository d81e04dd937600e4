cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05w; mkdir -p $O; L=$PWD/rsem_amd
( timeout 200 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > $O/product.log 2>&1 ); grep "ms/round" $O/product.log
for v in gxsl4 gxsl8 gxsl16 gxsl32; do ( RSEM_HIP_LIB=$L/librsem_hip_$v.so timeout 200 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > $O/$v.log 2>&1 ); echo "$v: $(grep 'ms/round' $O/$v.log)"; done
