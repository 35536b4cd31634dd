"""`python neddf/scripts/run_eval.py <run_dir> [--epoch N]` -- same command line,
inputs (`<run_dir>/.hydra/config.yaml`, `<run_dir>/models/model_{epoch:05}.pth`)
and outputs (`<run_dir>/eval/*.png`, psnr/ssim printout) as the reference's
neddf/scripts/run_eval.py:10-44.  The frozen config is read with PyYAML and the
`dataset.data_split=test` override applied by hand (hydra is not required).

Under a launcher (`python -m torch.distributed.run --nproc-per-node N neddf/scripts/run_eval.py <run_dir>`; not in the
reference) the rays of every view are sharded over the N GPUs (contiguous slabs of the pixel index, one RCCL all-gather of
20 B/ray per view, BASELINE.json configs[3]); rank 0 writes the PNGs and prints the metrics.  Every rank seeds identically
and jumps torch's CPU generator to its slab, so the images are the single-GPU ones bit for bit."""
import os
from argparse import ArgumentParser
from pathlib import Path

import torch
import yaml

from neddf_amd.config import instantiate


def main(argv=None) -> None:
    parser = ArgumentParser()
    parser.add_argument("output_dir", type=Path, help="directory path where models and render are located")
    parser.add_argument("--epoch", type=int, default=2000, help="epoch number of model")
    parser.add_argument("--seed", type=int, default=None,
                        help="torch seed for the sample uniforms (not a reference option: the reference's run_eval never seeds, and "
                             "torch seeds its default generator randomly per process)")
    args = parser.parse_args(argv)
    output_dir = args.output_dir.resolve()
    conf = output_dir / ".hydra" / "config.yaml"
    assert conf.is_file(), conf
    cfg = yaml.safe_load(open(conf))
    cfg["dataset"]["data_split"] = "test"
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        n_dev = torch.cuda.device_count()
        # NEDDF_DIST_BACKEND=gloo: ranks may share devices (tests on a one-GPU box); the pixel slabs are then staged through the host
        backend = os.environ.get("NEDDF_DIST_BACKEND", "nccl")
        local = local % max(n_dev, 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
        cfg["trainer"]["device"] = "cuda:%d" % local
    trainer = instantiate(cfg["trainer"], global_config=cfg, _recursive_=False)
    trainer.writes_outputs = rank == 0
    trainer.load_pretrained_model(output_dir / "models/model_{:05}.pth".format(args.epoch))
    save_dir = args.output_dir / "eval"
    if rank == 0:
        save_dir.mkdir(exist_ok=True)
    # ray sharding replays ONE stream of uniforms (each rank jumps the CPU generator to its slab): every rank needs rank 0's seed
    seed = args.seed
    if world > 1:
        box = [torch.initial_seed() if seed is None else seed]
        torch.distributed.broadcast_object_list(box, src=0)
        seed = int(box[0]) % (1 << 63)
    if seed is not None:
        torch.manual_seed(seed)
    trainer.render_all(save_dir)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
