#!/usr/bin/env python3
"""Binary ResNet-18 inference with the recipe of the reference's examples/cifar10.py:61-71 (XNOR weights, sign
activations, first and last layer real-valued) on one MI355X:

  1. drop-in      model(x)                     the reference's own call: the fused executor by itself (AutoFusion: stem
                                               launch on the caller's tensor + HIP graph of the rest, two halves in flight)
  2. per layer    per_layer_forward()          every binary layer = one launch, every BatchNorm (+ add) (+ ReLU) tail one
                                               launch, stem and head their kernels; library_tails(): torch's own modules
  3. fused        FusedResNet(model)(x)        the executor, explicitly, eager launches
  4. pipelined    PipelinedInference(...)      HIP-graph replay, two batches in flight

    python examples/infer_resnet18.py [--batch 256] [--checkpoint model.bnnpack]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "binary-networks-pytorch_amd")]

import torch  # noqa: E402

import bnn_amd as bnn  # noqa: E402
from bnn_amd import checkpoint  # noqa: E402
from bnn_amd.inference import FusedResNet, PipelinedInference, library_tails, per_layer_forward  # noqa: E402
from bnn_amd.models import resnet18  # noqa: E402
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--checkpoint", help="BNNPACK1 file written by bnn_amd.checkpoint.save_packed")
    args = ap.parse_args()
    dev = torch.device("cuda:0")

    bconfig = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                          weight_pre_process=XNORWeightBinarizer)
    model = bnn.prepare_binary_model(resnet18(), bconfig,
                                     custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    if args.checkpoint:
        model.load_state_dict(checkpoint.load_packed(args.checkpoint))
    else:
        # no trained weights at hand: give BatchNorm plausible statistics.  (The default initialisation zeroes
        # bn2.weight of every block, so all residual branches are exactly 0 and exact ties decide the signs.)
        g = torch.Generator().manual_seed(0)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data = torch.rand(m.num_features, generator=g) + 0.5
                m.bias.data = torch.randn(m.num_features, generator=g) * 0.3
                m.running_mean = torch.randn(m.num_features, generator=g) * 0.5
                m.running_var = torch.rand(m.num_features, generator=g) + 0.5
    model = model.to(dev).eval()
    x = torch.randn(args.batch, 3, 224, 224, device=dev)

    def rate(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return args.batch * n / (time.perf_counter() - t0)

    with torch.no_grad():
        with per_layer_forward():
            y_layer = model(x)
            print("per layer  %9.0f images/s" % rate(lambda: model(x), 5))
            with library_tails():                    # the reference's own formulation: torch stem / BatchNorm / ReLU / add
                y_ref = model(x)
                print("  (library tails %9.0f images/s)" % rate(lambda: model(x), 3))
        xs = [x, x + 0.01, x - 0.01]                 # a NEW tensor every call, as an eval loop delivers them
        it = iter(range(10 ** 9))
        print("drop-in    %9.0f images/s   (model(x), a new tensor every call)" % rate(
            lambda: model(xs[next(it) % 3]), 20))
        fused = FusedResNet(model)
        assert torch.equal(model(x), fused(x))       # the call and the explicit executor: same bits
        assert torch.equal(y_layer, fused(x))        # ... and the per-layer path with its one-launch tails
        # folded BatchNorm (one fma) and torch's BatchNorm round differently; an activation that lands within an
        # ulp of 0 can therefore binarise differently and move that image's logits — a handful per thousand
        close = ((fused(x) - y_ref).abs().amax(dim=1) <= 1e-3 * y_ref.abs().max()).float().mean().item()
        assert close > 0.95, close
        print("fused == library composition to 1e-3 on %.1f %% of the images" % (100 * close))
        print("fused      %9.0f images/s" % rate(lambda: fused(x), 20))
        pipe = PipelinedInference(model, x)
        it = iter(range(10 ** 9))
        print("pipelined  %9.0f images/s" % rate(lambda: pipe.launch(next(it)), 40))
        pipe.synchronize()


if __name__ == "__main__":
    main()
