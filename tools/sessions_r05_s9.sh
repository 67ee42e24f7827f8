mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r05/collect_profiles:900:bash tools/collect_profiles.sh r05" \
 "r05/gputests_final3:1100:python3 -m pytest tests -m gpu -x -q --durations=6" \
 "r05/smoke:300:python3 -c 'import __graft_entry__ as g; g.smoke()'" \
 "r05/bench_driver_style:300:python3 bench.py --gpus 1 --steps 20 --warmup 5" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session9_summary.txt
