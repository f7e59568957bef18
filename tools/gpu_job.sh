set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd ab/hist_nocopy_pkscalar && timeout 900 python -m pytest tests/test_fullsize_gpu.py -k "deterministic_at_scale" -q 2>&1 | tail -6; timeout 900 python -m pytest tests/test_backward_parity.py -m gpu -q 2>&1 | tail -4) > gpurun_out/hist_nocopy_pkscalar.txt 2>&1
cat gpurun_out/hist_nocopy_pkscalar.txt
(cd ab/hist_nocopy && timeout 900 python -m pytest tests/test_backward_parity.py -m gpu -q 2>&1 | tail -4) 2>&1 | tail -5
