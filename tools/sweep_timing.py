import sys, json
sys.path.insert(0, '.')
import bench
for i in range(4):
    r = bench.sigma_sweep_config0()
    print(json.dumps({k: round(r[k], 4) for k in ('wall_s', 'create_task_s', 'train_s', 'validate_s', 'test_s')}), flush=True)
