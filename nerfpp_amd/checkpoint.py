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
    """NeRFExecutor.h:540-566: whichever of the four module files exist + the start step + the Adam state; `would_restore` is the reference's own
    existence condition (:541-546) -- when False its executor ignores the directory and calls Initialize()."""
    out = {}
    for key, fn in (("embedder", "embedder_checkpoint.pt"), ("model", "model_checkpoint.pt"), ("lang_embedder", "lang_embedder_checkpoint.pt"),
                    ("lang_model", "lang_model_checkpoint.pt")):
        f = os.path.join(path, fn)
        if os.path.exists(f):
            out[key], out[key + "_buffers"] = load_module(f)
    f = os.path.join(path, "start_checkpoint.pt")
    if os.path.exists(f):
        out["start"] = int(load_tensor(f).reshape(-1)[0])
    f = os.path.join(path, "optimizer_checkpoint.pt")
    if os.path.exists(f):
        out["optimizer"] = load_adam(f)
    out["would_restore"] = WouldRestore(path, use_nerf=True, use_lerf=("lang_model" in out))
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


# ------------------------------------------------------------------------------------------------
# torch::optim::Adam archives (torch::save(*Optimizer, "optimizer_checkpoint.pt"), NeRFExecutor.h:1067; restored at :565)
#
# LibTorch's optimizer serialisation (torch/csrc/api/include/torch/optim/serialize.h) writes a module archive
#     pytorch_version = "1.5.0"
#     state/<key>/{step (int), exp_avg, exp_avg_sq}              one sub-archive per parameter THAT HAS STATE, <key> = a decimal number
#     param_groups/{param_groups/size, param_groups/<g>/{params/size, params/<i> = <key>, options/{lr, betas, eps, weight_decay, amsgrad}}}
# and on load hands state <key> to the optimizer's i-th parameter of group g -- the keys themselves (addresses in the writing process)
# only have to be distinct numbers.  The attribute names are not Python identifiers, so the archive is assembled with the same low-level
# module builder TorchScript uses instead of torch.jit.script.
# ------------------------------------------------------------------------------------------------
class _Archive(torch.nn.Module):
    pass


def _archive(attrs, subs=()):
    """OutputArchive::write for a list of (name, tensor | int | float | bool | str | tuple) and sub-archives (name, archive)."""
    b = torch._C.ConcreteModuleTypeBuilder(_Archive)
    for name, v in attrs:
        if torch.is_tensor(v):
            b.add_attribute(name, torch._C.TensorType.get(), True, False)
        else:
            b.add_attribute(name, torch._C._jit_try_infer_type(v).type(), False, False)
    for name, (ct, _) in subs:
        b.add_module(name, ct)
    ct = b.build()
    cm = torch._C._create_module_with_type(ct.jit_type)
    for name, v in attrs:
        cm.setattr(name, v)
    for name, (_, sub) in subs:
        cm.setattr(name, sub)
    return ct, cm


def save_adam(path, moments, step, lr, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0, amsgrad=False):
    """Write an archive torch::load(torch::optim::Adam&) accepts.  moments: per parameter, in the optimizer's parameter order (NeRFExecutor.h:508-535:
    embedder parameters, then the model's), a pair (exp_avg, exp_avg_sq) of arrays shaped like the parameter, or None for a parameter without state."""
    keys = [str(1000 + i) for i in range(len(moments))]
    states = []
    for k, mv in zip(keys, moments):
        if mv is None:
            continue
        m, v = (torch.as_tensor(np.ascontiguousarray(a, np.float32)) for a in mv)
        states.append((k, _archive([("step", int(step)), ("exp_avg", m), ("exp_avg_sq", v)])))
    opts = _archive([("lr", float(lr)), ("betas", (float(betas[0]), float(betas[1]))), ("eps", float(eps)), ("weight_decay", float(weight_decay)), ("amsgrad", bool(amsgrad))])
    g0 = _archive([("params/size", torch.tensor(len(keys)))] + [(f"params/{i}", k) for i, k in enumerate(keys)], [("options", opts)])
    groups = _archive([("param_groups/size", torch.tensor(1))], [("param_groups/0", g0)])
    top = _archive([("pytorch_version", "1.5.0")], [("state", _archive([], states)), ("param_groups", groups)])
    top[1].save(path)


def load_adam(path):
    """-> dict(step, lr, betas, eps, moments = [(exp_avg, exp_avg_sq) | None per parameter of group 0, in order])."""
    c = torch.jit.load(path, map_location="cpu")._c
    state, g0 = c.getattr("state"), c.getattr("param_groups").getattr("param_groups/0")
    n = int(g0.getattr("params/size"))
    moments, step = [], 0
    for i in range(n):
        k = g0.getattr(f"params/{i}")
        if state.hasattr(k):
            st = state.getattr(k)
            moments.append((st.getattr("exp_avg").detach().numpy().copy(), st.getattr("exp_avg_sq").detach().numpy().copy()))
            step = int(st.getattr("step"))
        else:
            moments.append(None)
    o = g0.getattr("options")
    return dict(step=step, lr=float(o.getattr("lr")), betas=tuple(o.getattr("betas")), eps=float(o.getattr("eps")), moments=moments)


def WouldRestore(path, use_nerf=True, use_lerf=False):
    """The existence condition under which NeRFExecutor::Initialize restores instead of initialising (NeRFExecutor.h:541-546)."""
    ex = lambda f: os.path.exists(os.path.join(path, f))
    return (ex("start_checkpoint.pt") and ex("optimizer_checkpoint.pt") and (ex("model_checkpoint.pt") or not use_nerf) and
            (ex("lang_embedder_checkpoint.pt") or not use_lerf))


def SaveCheckpoint(path, embedder=None, embedder_buffers=None, model=None, global_step=0, optimizer=None):
    """NeRFExecutor.h:1055-1070: embedder + model + start step + optimizer.  `optimizer`: dict(moments, step, lr[, betas, eps]) for save_adam -- without
    optimizer_checkpoint.pt the reference's executor ignores the directory and initialises afresh (:541-546)."""
    os.makedirs(path, exist_ok=True)
    if embedder is not None:
        save_module(os.path.join(path, "embedder_checkpoint.pt"), embedder, embedder_buffers)
    if model is not None:
        save_module(os.path.join(path, "model_checkpoint.pt"), model)
    save_tensor(os.path.join(path, "start_checkpoint.pt"), np.full((1,), int(global_step), np.int64))
    if optimizer is not None:
        save_adam(os.path.join(path, "optimizer_checkpoint.pt"), **optimizer)
