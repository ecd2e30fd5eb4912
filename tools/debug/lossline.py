import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["n_gpus"], d["config"]["per_gpu_batch"], round(d["ms_per_step"], 3), "loss/sample", d["loss_per_sample"])
