# round 5: the whole GPU suite with durations
mkdir -p gpurun_out/r5w
timeout 2400 python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/r5w/tests.log 2>&1
tail -50 gpurun_out/r5w/tests.log
