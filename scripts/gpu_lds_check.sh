cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
for W in 0 1; do echo "MFX_WARM_L2=$W"; MFX_WARM_L2=$W python scripts/gpu_ldmode.py lib 2>&1 | grep -v amdgpu; done
