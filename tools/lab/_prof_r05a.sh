bash tools/profile_round.sh r05_a > gpurun_out/prof_r05a.log 2>&1; tail -3 gpurun_out/prof_r05a.log | cut -c1-600
