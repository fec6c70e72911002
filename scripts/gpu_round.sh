set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; echo "bench rc=$?"
cat gpurun_out/bench_r01.json
