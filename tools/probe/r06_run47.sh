cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for c in cfg2 cfg3 cfg5; do
  bash tools/pmc_a5.sh d5f9111 r06/pmc_final_$c $c hbm > gpurun_out/r06/pmc_final_$c.log 2>&1
done
