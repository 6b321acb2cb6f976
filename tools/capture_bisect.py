#!/usr/bin/env python3
"""Which piece of the step refuses hipGraph capture? Captures each piece separately and reports."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib  # noqa: E402
from adaptiveisp_amd.agent import Agent, one_hot, pdf_sample  # noqa: E402
from adaptiveisp_amd.config import cfg  # noqa: E402
from adaptiveisp_amd.util import enrich_image_input  # noqa: E402
from adaptiveisp_amd.yolo import YoloEngine, yolov3  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.enabled = "--miopen" in sys.argv
B, H, W = 2, 96, 160
x = torch.rand(B, 3, H, W, device=dev)
agent = Agent(cfg, shape=(16, 64, 64), device=dev).to(dev).eval()
z = torch.rand(B, cfg.z_dim, device=dev)
s0 = torch.zeros(B, 13, device=dev)
eng = YoloEngine(yolov3().eval(), B, H, W, device=dev)
ids = torch.tensor([0, 4], dtype=torch.int32, device=dev)
pp = torch.rand(B, 24, device=dev)
pooled = _lib.pool64(x)
net_in = enrich_image_input(cfg, pooled, s0)


def pieces():
    yield "pool64", lambda: _lib.pool64(x)
    yield "isp_forward", lambda: _lib.forward(x, ids, pp, clip=True)
    yield "engine", lambda: eng(x)
    yield "enrich", lambda: enrich_image_input(cfg, pooled, s0)
    yield "trunk", lambda: agent.feature_extractor(net_in)
    feats = agent.feature_extractor(net_in)
    yield "heads", lambda: [f.filter_param_regressor(f.extract_parameters(feats)[0]) for f in agent.filters]
    yield "selector", lambda: agent.softmax(agent.fc2(agent.lrelu(agent.fc1(agent.action_selection(net_in)))))
    pdf = agent.softmax(agent.fc2(agent.lrelu(agent.fc1(agent.action_selection(net_in)))))
    yield "pdf_sample", lambda: pdf_sample(pdf, z[:, 0:1])
    yield "argmax", lambda: torch.argmax(pdf, dim=1).to(torch.int32)
    sel = torch.full((B,), 3, dtype=torch.int64, device=dev)
    yield "full", lambda: torch.full((B,), 3, dtype=torch.int64, device=dev)
    yield "one_hot", lambda: one_hot(10, sel)
    yield "op_ids", lambda: agent._op_ids(sel)
    params = [f.filter_param_regressor(f.extract_parameters(feats)[0]) for f in agent.filters]
    yield "packed", lambda: agent._packed_params(params, sel)
    yield "get_mask", lambda: agent.filters[0].get_mask(x)
    yield "agent_full", lambda: agent((x, z, s0), 1.0, selected_filter_id=3)


with torch.no_grad():
    for name, fn in pieces():
        fn()
        torch.cuda.synchronize()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            g.replay()
            torch.cuda.synchronize()
            print(f"{name:12s} capture OK", flush=True)
        except Exception as e:
            print(f"{name:12s} capture FAILED: {str(e).splitlines()[0][:100]}", flush=True)
            torch.cuda.synchronize()
