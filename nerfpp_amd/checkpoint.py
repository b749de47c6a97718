"""Checkpoint interchange with the reference -- NeRFExecutor::SaveCheckpoint / LoadCheckpoint (NeRFExecutor.h:540-566, :1055-1070),
SURVEY section 8f row N3.

The reference writes `torch::save(module, path)` archives: embedder_checkpoint.pt, model_checkpoint.pt, (lang_*), start_checkpoint.pt.
Those are TorchScript module archives, so `torch.jit.load` reads them and a scripted nn.Module with the same parameter / buffer names
writes one that `torch::load(module, path)` accepts.  Host-side I/O only: no GPU work here.

Parameter names (what `named_parameters()` yields in the reference and therefore the order of the C ABI's parameter blobs):
    NeRFSmall      model_sigma_net_{l}.weight ..., model_color_net_{l}.weight ...                       (NeRF.cpp:350-354)
    NeRF           model_pts_linears_{i}.{weight,bias} ..., views_linears / feature / alpha / rgb ...    (NeRF.cpp:63-88)
    HashEmbedder   embedder_embeddings_{l}.weight  [2^T, F]                                            (NeRF.cpp:255-256)
    CuHashEmbedder embedder_embeddings [L*2^T, F] + buffers embedder_primes [L,1,3] int32, embedder_biases [L,3],
                   embedder_feat_local_size [L], embedder_feat_local_idx [L]                            (CuHashEmbedder.cpp:24,73-76)
"""
import os
from collections import OrderedDict

import numpy as np
import torch


def load_module(path):
    """-> (OrderedDict name -> np.ndarray of parameters, OrderedDict of buffers), in the archive's (= named_parameters()) order."""
    m = torch.jit.load(path, map_location="cpu")
    params = OrderedDict((k, v.detach().numpy().copy()) for k, v in m.named_parameters())
    bufs = OrderedDict((k, v.detach().numpy().copy()) for k, v in m.named_buffers())
    return params, bufs


def load_tensor(path):
    """torch::save(tensor, path): an archive holding the tensor under the key "0"."""
    m = torch.jit.load(path, map_location="cpu")
    return dict(m.named_parameters(), **dict(m.named_buffers()))["0"].detach().numpy().copy()


def blob(params):
    """Concatenate in checkpoint order: the `params` argument of nrf_mlp_*_create / the table of nrf_hash_set_table (NGP mode)."""
    return np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in params.values()])


def LoadCheckpoint(path):
    """NeRFExecutor.h:540-566: whichever of the four module files exist + the start step."""
    out = {}
    for key, fn in (("embedder", "embedder_checkpoint.pt"), ("model", "model_checkpoint.pt"), ("lang_embedder", "lang_embedder_checkpoint.pt"),
                    ("lang_model", "lang_model_checkpoint.pt")):
        f = os.path.join(path, fn)
        if os.path.exists(f):
            out[key], out[key + "_buffers"] = load_module(f)
    f = os.path.join(path, "start_checkpoint.pt")
    if os.path.exists(f):
        out["start"] = int(load_tensor(f).reshape(-1)[0])
    return out


def cu_hash_state(params, bufs, module_name="embedder"):
    """CuHashEmbedder checkpoint -> (table [L*2^T, F] fp32, primes [L*3] int32, biases [L,3])."""
    table = params[module_name + "_embeddings"]
    primes = bufs[module_name + "_primes"].reshape(-1).astype(np.int32)
    biases = bufs[module_name + "_biases"].reshape(-1, 3).astype(np.float32)
    return table, primes, biases


class _Holder(torch.nn.Module):
    def forward(self):          # never called; a scripted module needs a method table
        return 0


def _build(params, bufs):
    root = _Holder()
    def put(name, t, is_param):
        mod, parts = root, name.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Holder())
            mod = getattr(mod, p)
        t = torch.as_tensor(np.ascontiguousarray(t))
        if is_param:
            mod.register_parameter(parts[-1], torch.nn.Parameter(t, requires_grad=t.is_floating_point()))
        else:
            mod.register_buffer(parts[-1], t)
    for k, v in params.items():
        put(k, v, True)
    for k, v in (bufs or {}).items():
        put(k, v, False)
    return root


def save_module(path, params, bufs=None):
    """Write a torch::load-able module archive carrying `params` (and `bufs`) under the given dotted names."""
    torch.jit.script(_build(params, bufs)).save(path)


def save_tensor(path, value):
    m = _Holder()
    m.register_buffer("0", torch.as_tensor(np.ascontiguousarray(value)))
    torch.jit.script(m).save(path)


def SaveCheckpoint(path, embedder=None, embedder_buffers=None, model=None, global_step=0):
    """NeRFExecutor.h:1055-1070 (embedder + model + start step; the optimizer state is not interchanged)."""
    os.makedirs(path, exist_ok=True)
    if embedder is not None:
        save_module(os.path.join(path, "embedder_checkpoint.pt"), embedder, embedder_buffers)
    if model is not None:
        save_module(os.path.join(path, "model_checkpoint.pt"), model)
    save_tensor(os.path.join(path, "start_checkpoint.pt"), np.full((1,), int(global_step), np.int64))
