timeout -k 10 600 python -m pytest tests/test_gpu_steps.py -q -m gpu -k "hgetf2" -x > gpurun_out/r04_z8_steps.log 2>&1 || { tail -30 gpurun_out/r04_z8_steps.log; exit 1; }
tail -2 gpurun_out/r04_z8_steps.log
timeout -k 10 300 python tools/hp_stamp_probe.py > gpurun_out/r04_z8_stamp.log 2>&1; cat gpurun_out/r04_z8_stamp.log
timeout -k 10 300 python tools/hp_window_probe.py > gpurun_out/r04_z8_probe.log 2>&1
head -9 gpurun_out/r04_z8_probe.log
