#!/usr/bin/env python3
"""HIP path vs the REFERENCE's own outputs on the trained-weights fixtures (tests/golden/*_trained_*.npz), key by key
(GPU box).  The tests assert bounds; this prints what the errors actually are, beside the same errors of the random-init
fixtures of the same shape, so that the bounds in tests/test_gpu_parity.py are chosen from data.
Usage: python tools/trained_fixture_report.py [--json profiles/r04_trained_fixture_report.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    import reflect_sampling_nerf_amd as pkg
    from tests.helpers import load_golden, max_abs

    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rep = {}

    def model_for(meta, g, train):
        s = meta["samples"]
        cfg = pkg.ReflectSamplingNeRFModelConfig(num_coarse_samples=s[0], num_importance_samples=s[1],
                                                num_reflect_coarse_samples=s[2], num_reflect_importance_samples=s[3],
                                                base_mlp_num_layers=meta["layers"], base_mlp_layer_width=meta["width"])
        m = cfg.setup(scene_box=None, num_train_data=1)
        m.field.load_state_dict(g["param"])
        m.to(dev)
        return m.train() if train else m.eval()

    for name in ("eval_l8_w256", "eval_trained_l8_w64", "eval_trained_l8_w256"):
        if not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz")):
            continue
        meta, g = load_golden(name)
        m = model_for(meta, g, False)
        i = g["in"]
        rb = pkg.RayBundle(**{k: i[k].to(dev) for k in ("origins", "directions", "pixel_area", "nears", "fars")})
        out = m(rb)
        ref = g["out"]
        r = {"R": meta["R"], "M": meta["M"], "mask_flips": int((out["mask"].cpu().to(torch.uint8) != ref["mask"]).sum())}
        for k, v in ref.items():
            if k == "mask" or k == "depth_reflect_fine":
                continue
            e = (out[k].cpu().double() - v.double()).abs()
            if "reflect" in k or k.startswith("depth"):
                per_ray = e.reshape(e.shape[0], -1).max(dim=1).values
                r[k] = {"max": float(e.max()), "rays_over_1e-4": int((per_ray > 1e-4).sum()), "median": float(per_ray.median())}
            else:
                r[k] = {"max": float(e.max())}
        if meta["M"] > 0:
            v = ref["depth_reflect_fine"]
            e = ((out["depth_reflect_fine"].cpu() - v).abs() / (1 + v.abs()))
            r["depth_reflect_fine_rel"] = {"max": float(e.max()), "rays_over_1e-4": int((e > 1e-4).sum())}
        rep[name] = r
        print(name, json.dumps(r), flush=True)

    for name in ("trainstep_l8_w256", "trainstep_trained_l8_w64", "trainstep_trained_l8_w256"):
        if not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz")):
            continue
        for inject, mma in ((True, "f32"), (False, "f32"), (True, "bf16x6")):
            meta, g = load_golden(name)
            m = model_for(meta, g, True)
            m.field.set_mma_mode(mma)
            i = g["in"]
            rb = pkg.RayBundle(**{k: i[k].to(dev) for k in ("origins", "directions", "pixel_area", "nears", "fars")})
            out = m._get_outputs_train(rb, jitter={k: v.to(dev) for k, v in g["jitter"].items()}, bins=g["bins"] if inject else None)
            ref = g["out"]
            r = {"mask_flips": int((out["mask"].cpu().to(torch.uint8) != ref["mask"]).sum()), "M": meta["M"]}
            for k, v in ref.items():
                if k in ("mask", "depth_reflect_fine"):
                    continue
                r[k] = float((out[k].detach().cpu().double() - v.double()).abs().max())
            checked = dict(out)
            checked["normals_coarse"], checked["normals_fine"] = ref["normals_coarse"].to(dev), ref["normals_fine"].to(dev)
            losses = m.get_loss_dict(checked, {"image": i["image"].to(dev)})
            r["loss_rel"] = {k: abs(float(losses[k].detach()) - float(v)) / max(abs(float(v)), 1e-3) for k, v in g["loss"].items()}
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            gr = {}
            for pn, p in m.field.named_parameters():
                if pn in g["grad"]:
                    a_, b_ = p.grad.cpu().flatten().double(), g["grad"][pn].flatten().double()
                    gr[pn] = {"err_over_max": float((a_ - b_).abs().max() / (b_.abs().max() + 1e-300)),
                              "rel_l2": float((a_ - b_).norm() / (b_.norm() + 1e-300)),
                              "cos": float(torch.dot(a_, b_) / (a_.norm() * b_.norm() + 1e-300))}
            r["worst_grad_err_over_max"] = max(v["err_over_max"] for v in gr.values())
            r["worst_grad_rel_l2"] = max(v["rel_l2"] for v in gr.values())
            r["worst_grad_cos"] = min(v["cos"] for v in gr.values())
            r["grads"] = gr
            key = f"{name}|inject_bins={inject}|{mma}"
            rep[key] = r
            print(key, json.dumps({k: v for k, v in r.items() if k != "grads"}), flush=True)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
