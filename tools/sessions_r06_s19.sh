mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/cycles_high_1:300:python3 tools/measure_cycles.py --reps 5 --best" \
 "r06/cycles_default_1:300:ANEMOI_SAMPLER_STREAMS=default python3 tools/measure_cycles.py --reps 5 --best" \
 "r06/cycles_high_2:300:python3 tools/measure_cycles.py --reps 5 --best" \
 "r06/cycles_default_2:300:ANEMOI_SAMPLER_STREAMS=default python3 tools/measure_cycles.py --reps 5 --best" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session19_summary.txt
