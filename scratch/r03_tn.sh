set -euo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_tn" 2>&1 | tail -3
for lib in default ns2 ns2a8; do
echo "== $lib"
if [ $lib = default ]; then unset DSKD_HIP_LIB || true; else export DSKD_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libs/libdskd_$lib.so; fi
timeout -k 10 600 python scratch/r03_tn.py 2>&1 | grep -v amdgpu
done
