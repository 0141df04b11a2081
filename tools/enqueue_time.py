"""Host-side enqueue time of the training step's parts (GPU box): forward until the host wait, loss, backward, optimiser.
Each part is timed from a synchronised device to the return of the Python call (the GPU runs behind).

    python tools/enqueue_time.py
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def main():
    import reflect_sampling_nerf_amd as pkg
    from reflect_sampling_nerf_amd.parallel import apply_loss_warmup
    from reflect_sampling_nerf_amd.synthetic import synthetic_rays

    pkg.load_library()
    dev = torch.device("cuda", 0)
    R = 4096
    torch.manual_seed(0)
    model = pkg.ReflectSamplingNeRFModelConfig().setup(scene_box=None, num_train_data=1)
    with torch.no_grad():
        model.field.field_output_density.net.bias += 2.0
    model.to(dev).train()
    o, d, pa = synthetic_rays(R, seed=0)
    rb = pkg.RayBundle(origins=o.to(dev), directions=d.to(dev), pixel_area=pa.reshape(R, 1).to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev))
    batch = {"image": torch.rand(R, 3).to(dev)}
    params = model.get_param_groups()["fields"]
    opt = pkg.FusedRAdam(params, lr=1e-3, eps=1e-15)
    acc = {"forward (incl. the wait for M)": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0, "gpu step": 0.0}
    n = 10
    for it in range(n + 3):
        apply_loss_warmup(model, 100 + it)
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(rb)
        t1 = time.perf_counter()
        loss = sum(model.get_loss_dict(out, batch).values())
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        if it >= 3:
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t0)):
                acc[k] += v
    for k, v in acc.items():
        print("%-34s %7.2f ms" % (k, v / n * 1e3))


if __name__ == "__main__":
    main()
