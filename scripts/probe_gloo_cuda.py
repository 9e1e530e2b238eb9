"""Probe (GPU box): can two gloo ranks sharing cuda:0 run the collectives of the training step on CUDA tensors?
(all_reduce, reduce_scatter_tensor, all_gather_into_tensor).  Decides how the -m gpu world-size-2 training test is written."""
import os, sys, subprocess, socket
if "RANK" not in os.environ:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.abspath(__file__)], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-3000:]); sys.exit(r.returncode)
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = "cuda:0"
for name, fn in (("all_reduce", lambda: dist.all_reduce(torch.ones(8, device=dev))),
                 ("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(torch.zeros(4, device=dev), torch.ones(8, device=dev))),
                 ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.zeros(8, device=dev), torch.ones(4, device=dev)))):
    try:
        fn(); torch.cuda.synchronize(); print(f"rank {rank}: gloo {name} on CUDA tensors: ok", flush=True)
    except Exception as e:
        print(f"rank {rank}: gloo {name} on CUDA tensors: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
dist.destroy_process_group()
