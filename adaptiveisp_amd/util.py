"""Host-side helpers of the hot path (reference: util.py:15-18,58-63,67-99; dataloader.py:401-418)."""
import numpy as np
import torch

# layout of the per-image state vector: [has-reward, stopped, step, usage flag per filter ...]
STATE_REWARD_DIM = 0
STATE_STOPPED_DIM = 1
STATE_STEP_DIM = 2
STATE_DROPOUT_BEGIN = 3


class Dict(dict):
    """dict with attribute access (reference util.py:67-99); cfg is one of these."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            if isinstance(a, dict):
                self.update(a)
        self.update(kwargs)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def enrich_image_input(cfg, net, states):
    """Append the state vector as constant planes: [B,C,h,w] + [B,S] -> [B,C+S,h,w] (util.py:58-63)."""
    if cfg.img_include_states:
        planes = states[:, :, None, None].expand(-1, -1, net.shape[2], net.shape[3])
        net = torch.cat([net, planes.to(net.dtype)], dim=1)
    return net


def get_noise(batch_size, z_type="uniform", z_dim=27):
    if z_type == "normal":
        return np.random.normal(0, 1, [batch_size, z_dim]).astype(np.float32)
    if z_type == "uniform":
        return np.random.uniform(0, 1, [batch_size, z_dim]).astype(np.float32)
    raise AssertionError("Unknown noise type: %s" % z_type)


def get_initial_states(batch_size, num_state_dim, filters_number):
    """All-zero states: nothing rewarded, nothing stopped, step 0, no filter used yet."""
    return np.zeros((batch_size, num_state_dim), dtype=np.float32)


def to_device_async(array, device, dtype=None):
    """A host array on `device` without draining the stream: staged in pinned memory (the caching host allocator keeps the
    block alive until the copy has run) and copied with non_blocking=True. A pageable source makes torch's copy synchronise
    the stream — i.e. wait for every kernel enqueued so far — which costs the RL loop its whole host / device overlap."""
    t = torch.as_tensor(array) if dtype is None else torch.as_tensor(array, dtype=dtype)
    if torch.device(device).type != "cuda" or t.is_cuda:            # already on a device: nothing to pin
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)
