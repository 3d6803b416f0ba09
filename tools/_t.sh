python -m pytest tests/test_gpu_host_layer.py tests/test_gpu_deterministic.py tests/test_gpu_cpp_model.py tests/test_gpu_sdf_default_config.py tests/test_gpu_reference_classes.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --step-trace gpurun_out/step_trace_sf.txt > /dev/null 2>&1
