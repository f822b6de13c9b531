cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pol
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC -Wno-unused-value tools/lds_polluter.hip -o tools/liblds_polluter.so 2>/dev/null
GAUSPCC_TEST_POLLUTE=tools/liblds_polluter.so GAUSPCC_TEST_POLLUTE_PATTERN=7fc00001 timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_conv_variants.py --deselect tests/test_gpu_robustness.py > gpurun_out/pol/pytest.log 2>&1; echo "rc $?" >> gpurun_out/pol/pytest.log
grep -v amdgpu.ids gpurun_out/pol/pytest.log | grep -E "passed|failed|FAILED|rc |Error|fault|Abort" | cut -c1-220 | head -30
