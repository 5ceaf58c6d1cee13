"""TEST INFRASTRUCTURE - build container only (imports the reference from /root/reference; never imported by the product path).

    python -m oracle.check_dropin            # prints one line per yaml + a JSON summary, exit code 1 on any failure

Proves INTEGRATION.md form A: the ```python block of "## A." is read OUT OF INTEGRATION.md and executed in the namespace of the
imported reference's `ultralytics/nn/tasks.py` - i.e. exactly what a maintainer who appends it to that file gets - and then every shipped
`cfg/models/v10/*.yaml` / `cfg/models/v10-3D/*.yaml` is built through the reference's OWN constructors
(`YOLOv10DetectionModel` / `YOLOv10_3DDetectionModel`, nn/tasks.py:283-318, 645-653), its own `parse_model`, its own host-side stride
probe (nn/tasks.py:300-310), `bias_init`, `initialize_weights`, `_predict_once` and `BaseModel.fuse()`.

Held against the UNPATCHED reference built from the same yaml before the block was applied:
  * every row of the model is a yolov10_3d_amd module (the rebinding took);
  * `state_dict()` keys and shapes, the `save` list, `model.stride` / head `.stride` (8 / 16 / 32; 8 / 16 for M-3D) are identical;
  * the deterministic `bias_init` values are identical (and equal tests/golden/bias_init.npz's constants where the widths match);
  * `load_state_dict(reference.state_dict(), strict=True)` succeeds (checkpoint interop, nn/tasks.py:258-260);
  * the reference's `_predict_once` over a host tensor yields the reference's own output shapes (train and eval mode) as META tensors;
  * `fuse()` folds every Conv to the reference's folded weight / bias and leaves < 10 BatchNorm layers (`is_fused`);
  * `init_criterion()` returns this package's loss class.
No arithmetic of the path runs here (there is no GPU in the build container): values are the GPU suite's business.
"""
from __future__ import annotations

import glob
import json
import os
import re
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import ref_shim as R  # noqa: E402


def integration_block() -> str:
    """the first ```python block under '## A.' of INTEGRATION.md, verbatim"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = text.index("## A.")
    m = re.search(r"```python\n(.*?)```", text[a:], re.S)
    assert m, "INTEGRATION.md: no python block under '## A.'"
    return m.group(1)


def yamls():
    base = os.path.join(R.REF_ROOT, "ultralytics", "cfg", "models")
    return sorted(glob.glob(os.path.join(base, "v10", "*.yaml")) + glob.glob(os.path.join(base, "v10-3D", "*.yaml")))


def _shapes(o):
    if torch.is_tensor(o):
        return tuple(o.shape)
    if isinstance(o, dict):
        return {k: _shapes(v) for k, v in o.items() if k in ("one2many", "one2one")}
    if isinstance(o, (list, tuple)):
        return [_shapes(v) for v in o]
    return None


def _det_biases(head):
    """the biases bias_init sets deterministically (head.py:95-109, 535-543, 847-871)"""
    out = {}
    if hasattr(head, "cls") and hasattr(head, "dep"):
        for name in ("cls", "o2d", "s2d", "o3d", "s3d", "dep"):
            for i, br in enumerate(getattr(head, name)):
                out[f"{name}/{i}"] = br[-1].bias.detach().clone()
    else:
        for name in ("cv2", "cv3", "one2one_cv2", "one2one_cv3"):
            for i, br in enumerate(getattr(head, name, [])):
                out[f"{name}/{i}"] = br[-1].bias.detach().clone()
    return out


def _folded(model):
    """{module path: (weight, bias)} of every fused Conv (anything with .conv carrying a bias and no .bn)"""
    out = {}
    for name, m in model.named_modules():
        c = getattr(m, "conv", None)
        if isinstance(c, torch.nn.Conv2d) and c.bias is not None and not hasattr(m, "bn"):
            out[name] = (c.weight.detach(), c.bias.detach())
    return out


def pristine(path):
    """summary of the unpatched reference built from `path` (3D yamls get the kernel-size default the reference lacks)"""
    rel = os.path.relpath(path, os.path.join(R.REF_ROOT, "ultralytics", "cfg", "models"))
    m = R.build_model(rel, seed=0)
    x = torch.zeros(1, 3, 256, 256)
    info = {
        "keys": {k: tuple(v.shape) for k, v in m.state_dict().items()},
        "state": {k: v.clone() for k, v in m.state_dict().items()},
        "save": list(m.save),
        "stride": m.stride.clone(),
        "bias": _det_biases(m.model[-1]),
        "train_shapes": _shapes(m.train()(x)),
    }
    m.load_state_dict(info["state"])  # the train-mode forward above moved the running statistics
    m.eval()
    with torch.no_grad():
        info["eval_shapes"] = _shapes(m(x))
        m.fuse(verbose=False)
    info["folded"] = _folded(m)
    info["is_fused"] = bool(m.is_fused())
    return info


def check(path, ref, T):
    import yolov10_3d_amd.loss as YL
    import yolov10_3d_amd.modules as YM

    errs = []
    is3d = "3D" in os.path.basename(path)
    cls = T.YOLOv10_3DDetectionModel if is3d else T.YOLOv10DetectionModel
    torch.manual_seed(0)
    m = cls(path, verbose=False)  # the reference's own yaml file, unchanged (no kernel-size shim)
    foreign = [type(r).__name__ for r in m.model if type(r).__module__ != YM.__name__ and not isinstance(r, torch.nn.Sequential)]
    foreign += [type(q).__name__ for r in m.model if isinstance(r, torch.nn.Sequential) for q in r if type(q).__module__ != YM.__name__]
    if foreign:
        errs.append(f"rows not rebound: {sorted(set(foreign))}")
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    if keys != ref["keys"]:
        only_r = sorted(set(ref["keys"]) - set(keys))[:4]
        only_m = sorted(set(keys) - set(ref["keys"]))[:4]
        diff = [k for k in keys if k in ref["keys"] and keys[k] != ref["keys"][k]][:4]
        errs.append(f"state_dict differs: only reference {only_r}, only drop-in {only_m}, shapes {diff}")
    if list(m.save) != ref["save"]:
        errs.append(f"save list {m.save} != {ref['save']}")
    want = [8.0, 16.0] if "yolov10m_3D" in path else [8.0, 16.0, 32.0]
    for name, s in (("model.stride", m.stride), ("head.stride", m.model[-1].stride)):
        if s.tolist() != want or not torch.equal(s, ref["stride"]):
            errs.append(f"{name} {s.tolist()} (reference {ref['stride'].tolist()}, expected {want})")
    bias = _det_biases(m.model[-1])
    bad = [k for k in ref["bias"] if k not in bias or not torch.equal(bias[k], ref["bias"][k])]
    if bad or not ref["bias"]:
        errs.append(f"bias_init differs at {bad[:4]}")
    try:
        m.load_state_dict(ref["state"], strict=True)
    except Exception as e:  # noqa: BLE001
        errs.append(f"load_state_dict(strict): {str(e)[:200]}")
    x = torch.zeros(1, 3, 256, 256)
    tr = m.train()(x)
    metas = [t for t in tr["one2many"]]
    if not all(t.device.type == "meta" for t in metas):
        errs.append("host forward produced non-meta tensors (something was computed on the host)")
    if _shapes(tr) != ref["train_shapes"]:
        errs.append(f"train-mode shapes {_shapes(tr)} != {ref['train_shapes']}")
    m.eval()
    ev = _shapes(m(x))
    if ev != ref["eval_shapes"]:
        errs.append(f"eval-mode shapes {ev} != {ref['eval_shapes']}")
    m.fuse(verbose=False)
    if not m.is_fused():
        errs.append("is_fused() is False after fuse()")
    fold = _folded(m)
    common = [k for k in fold if k in ref["folded"]]
    nconv = sum(1 for q in m.modules() if isinstance(q, YM.Conv))
    # RepVGGDW is not rebound: its two depth-wise Convs are folded one by one here, into ONE padded 7x7 in the reference
    if len(common) < 0.9 * len(ref["folded"]) or len(fold) != nconv:
        errs.append(f"fuse(): {len(fold)} folded of {nconv} Convs, {len(common)} in common with the reference's {len(ref['folded'])}")
    for k in common:
        if k.endswith(".conv") and any(k2 == k[: -len(".conv")] + ".conv1" for k2 in fold):
            continue  # RepVGGDW.conv: the reference's holds 7x7 + padded 3x3, ours the 7x7 alone (checked as a sum below)
        w, b = fold[k]
        rw, rb = ref["folded"][k]
        if w.shape != rw.shape or not torch.allclose(w, rw, rtol=1e-5, atol=1e-6) or not torch.allclose(b, rb, rtol=1e-5, atol=1e-6):
            errs.append(f"fuse(): folded weight / bias of {k} differ from the reference's")
            break
    for k in fold:  # RepVGGDW pairs: conv (7x7) + pad(conv1 (3x3)) == the reference's single 7x7
        if k.endswith(".conv1") and k[: -1] in fold and k[: -1] in ref["folded"]:
            w7, b7 = fold[k[: -1]]
            w3, b3 = fold[k]
            rw, rb = ref["folded"][k[: -1]]
            if not (torch.allclose(w7 + torch.nn.functional.pad(w3, [2, 2, 2, 2]), rw, rtol=1e-5, atol=1e-6) and torch.allclose(b7 + b3, rb, rtol=1e-5, atol=1e-6)):
                errs.append(f"fuse(): RepVGGDW pair {k[:-1]} does not sum to the reference's folded 7x7")
                break
    if _shapes(m(x)) != ref["eval_shapes"]:
        errs.append("eval-mode shapes changed after fuse()")
    m.args = R.model_args()
    crit = m.init_criterion()
    if type(crit).__module__ != YL.__name__:
        errs.append(f"init_criterion() -> {type(crit).__module__}.{type(crit).__name__}")
    return errs, {"rows": len(m.model), "keys": len(keys), "stride": m.stride.tolist(), "folded": len(fold), "params": sum(p.numel() for p in m.parameters())}


def main(quiet=False):
    R.import_reference()
    import ultralytics.nn.tasks as T

    ys = yamls()
    assert len(ys) == 12, ys
    refs = {p: pristine(p) for p in ys}  # before the block rebinds anything
    exec(compile(integration_block(), "INTEGRATION.md#A", "exec"), T.__dict__)
    summary, failed = {}, 0
    for p in ys:
        errs, info = check(p, refs[p], T)
        name = os.path.relpath(p, os.path.join(R.REF_ROOT, "ultralytics", "cfg", "models"))
        summary[name] = {"ok": not errs, "errors": errs, **info}
        failed += bool(errs)
        if not quiet:
            print(f"{'ok  ' if not errs else 'FAIL'} {name}: rows {info['rows']}, keys {info['keys']}, stride {info['stride']}, folded {info['folded']}"
                  + ("".join("\n     " + e for e in errs)))
    if not quiet:
        print(json.dumps({"yamls": len(ys), "failed": failed}))
    return summary


if __name__ == "__main__":
    s = main()
    sys.exit(1 if any(not v["ok"] for v in s.values()) else 0)
