export TESTS="tests/test_gpu_train_step.py tests/test_gpu_train_fused.py tests/test_gpu_tape.py"
export TAIL=30
export CMDS='python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 50 --train-loss l2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d[\"train_step\"]))"
NLT_BWD_OBS_STREAM=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 50 --train-loss l2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d[\"train_step\"]))"'
bash tools/gpu_call.sh
