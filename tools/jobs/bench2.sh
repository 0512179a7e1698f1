# bench.py under the driver's launcher with two ranks on ONE GPU (development mode: the ranks share the device, the exchange is the host one):
#   gpurun -- 'bash tools/jobs/bench2.sh'
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --frames 20000 --allow-host-exchange --no-extras --no-cpu-baseline 2>&1 | tail -5 | cut -c1-1500
