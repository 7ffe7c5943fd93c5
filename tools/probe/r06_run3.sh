set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r06/gpu_tests3.log
for v in default ffn1_2 bf16x3; do
  case $v in
    default) E="";;
    ffn1_2) E="LPM_DW_TERMS_FFN1=2";;
    bf16x3) E="LPM_DENSE_ARITHMETIC=bf16x3";;
  esac
  env $E timeout 600 python -m pytest "tests/test_gpu_models.py::test_untouched_reference_initialisation[cfg2]" -x -q -s 2>&1 | grep -E "^\[NetVladV1|passed|failed|Error" > gpurun_out/r06/untouched_cfg2_$v.log
done
timeout 600 python bench.py > gpurun_out/r06/bench3.json 2> gpurun_out/r06/bench3.err
LPM_DW_TERMS_FFN1=2 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench3_ffn1_2.json 2> gpurun_out/r06/bench3_ffn1_2.err
LPM_MHA_BWD_TERMS=3 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench3_mha3.json 2> gpurun_out/r06/bench3_mha3.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench3_b.json 2> gpurun_out/r06/bench3_b.err
