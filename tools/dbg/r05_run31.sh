mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_capi_host_gpu.py -q -m gpu > gpurun_out/r31_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r31_pytest.log
python tools/dbg/small_batch_encode.py > gpurun_out/small_batch_encode.log 2>&1
bash tools/dbg/ab_opt.sh "CONVDR_OPT_FFN2_SPLITK=0 CONVDR_OPT_FFN2_SPLITK=1" 4 > gpurun_out/ab_ffn2_splitk.log 2>&1
