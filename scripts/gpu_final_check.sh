cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4
bash scripts/gpu_multi_rehearsal.sh 2>&1 | grep "^N="
