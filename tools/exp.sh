timeout 600 python tools/shard_joint_probe.py 2 50000000 2>&1 | grep "ranks x" 
timeout 600 python tools/shard_joint_probe.py 4 25000000 2>&1 | grep "ranks x"
