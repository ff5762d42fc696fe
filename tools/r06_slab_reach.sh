set -u
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_slab.py -x -q -m gpu -k "jumps_of_several or flights or moving_cuts or matches_single_domain_oracle or rank_local_failure or several_ranks" 2>&1 | tail -15) > gpurun_out/r06_slab_reach_tests.txt
cat gpurun_out/r06_slab_reach_tests.txt
