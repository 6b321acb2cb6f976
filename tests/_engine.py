"""Test-only helpers: run the host-side Agent/Value logic on CPU tensors with the ORACLE standing in
for the HIP kernels (the product itself has no CPU path). Used by the `-m "not gpu"` tests."""
import numpy as np
import torch
import torch.nn as nn

import oracle


class OraclePool64(nn.Module):
    def forward(self, x):
        return torch.from_numpy(oracle.pool64(x.detach().cpu().numpy()))


def oracle_apply_isp(img, packed, op_ids):
    out = oracle.forward(img.detach().cpu().numpy(), op_ids.cpu().numpy().astype(np.int32),
                         packed.detach().cpu().numpy(), clip=True)
    return torch.from_numpy(out)


def cpu_agent(cfg, seed=0):
    from adaptiveisp_amd.agent import Agent
    from _synth import synth_state_dict
    ag = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device="cpu")
    ag.load_state_dict(synth_state_dict(ag, seed=seed))
    ag.down_sample = OraclePool64()
    ag._apply_isp = oracle_apply_isp
    return ag.eval()


def cpu_value(cfg, seed=1):
    from adaptiveisp_amd.value import Value
    from _synth import synth_state_dict
    va = Value(cfg, shape=(9 + len(cfg.filters), 64, 64))
    va.load_state_dict(synth_state_dict(va, seed=seed))
    va.down_sample = OraclePool64()
    return va.eval()
