"""Does `net(x)` (two halves on two internal streams) depend on how many OTHER streams the process made first?
HIP maps streams onto a few hardware queues; two streams that share one run their work back to back.
   python tools/exp_stream_queues.py [K ...]      (each K in its own process: the mapping is per process)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]


def one(k: int) -> None:
    import torch
    import bnn_amd as bnn
    from bnn_amd.models import resnet18
    from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
    dev = torch.device("cuda:0")
    busy = [torch.cuda.Stream(device=dev) for _ in range(k)]
    a = torch.ones(1 << 20, device=dev)
    for s in busy:                       # every stream has been used (queues are assigned on first use)
        with torch.cuda.stream(s):
            a = a * 1.0
    torch.cuda.synchronize()
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(resnet18(), cfg, custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    net = net.to(dev).eval()
    xs = [torch.randn(256, 3, 224, 224, device=dev) for _ in range(3)]
    with torch.no_grad():
        for i in range(300):
            net(xs[i % 3])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            net(xs[i % 3])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    from bnn_amd import inference
    print("other streams %2d: net(x) %.3f ms  (%.0f k images/s)  GPU_MAX_HW_QUEUES=%s  probe ratios %s" % (
        k, dt * 1e3, 256 / dt / 1e3, os.environ.get("GPU_MAX_HW_QUEUES", "-"), inference.STREAM_PROBE_LOG), flush=True)


if __name__ == "__main__":
    if os.environ.get("_ONE") is not None:
        one(int(os.environ["_ONE"]))
    else:
        for k in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 8, 11]:
            subprocess.run([sys.executable, __file__], env=dict(os.environ, _ONE=str(k)), check=False)
