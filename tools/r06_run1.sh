set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -k "attention or two_gpus or decoder_from_features or head_debug or rccl" > gpurun_out/r06_attn_tests.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_attn_tests.log | tail -15
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/attn_asm/redo_profile.py > gpurun_out/r06_attention_redo.txt 2>&1
rc=$?
tail -5 gpurun_out/r06_attention_redo.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/gpu_diag.py --only attention > gpurun_out/r06_diag_attention.log 2>&1
rc=$?
tail -5 gpurun_out/r06_diag_attention.log
exit $rc
