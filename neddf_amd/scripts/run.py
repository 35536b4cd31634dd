"""`python neddf/scripts/run.py [group=name ...] [a.b.c=value ...]` -- training entry with the command line of the
reference's neddf/scripts/run.py:13-40 (a hydra app over config/): the config groups of config/config.yaml's defaults
list are composed with PyYAML, `group=name` picks another file of a group, dotted keys override single values, and --
as hydra does -- the run executes inside a fresh outputs/<date>/<time>/ directory holding .hydra/config.yaml (which is
what run_eval.py later reads), models/, render/ and log/."""
import datetime
import os
import random
import sys
from pathlib import Path
from typing import Any, Dict, List

import numpy as np
import torch
import yaml

from neddf_amd.config import instantiate

CONFIG_DIR = Path(__file__).resolve().parents[2] / "config"


def compose(overrides: List[str], config_dir: Path = CONFIG_DIR) -> Dict[str, Any]:
    root = yaml.safe_load(open(config_dir / "config.yaml"))
    groups = {}
    for entry in root.get("defaults", []):
        (group, name), = entry.items()
        groups[group] = name
    values = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        if not _:
            raise SystemExit("override must be key=value: %r" % ov)
        if key in groups and "." not in key:
            groups[key] = val
        else:
            values.append((key, yaml.safe_load(val)))
    cfg: Dict[str, Any] = {}
    for group, name in groups.items():
        path = config_dir / group / (name + ".yaml")
        if not path.is_file():
            raise SystemExit("no config %s" % path)
        cfg[group] = yaml.safe_load(open(path))
    for key, val in values:
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = val
    return cfg


def main(argv=None) -> None:
    """Single process: the reference's behaviour.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N
    neddf/scripts/run.py ...`; not in the reference, which trains on one device) training is data-parallel over rays: rank r
    takes device cuda:r; the trainer (and with it the networks' initial weights) is built under the COMMON seed, the
    parameters are then broadcast from rank 0 and checked to be identical on every rank, and only after that each rank
    switches to seed 3408 + r (its own cameras / pixels); the gradients are averaged with one all-reduce per step
    (parallel.average_gradients), and only rank 0 creates the run directory and writes checkpoints, renders and logs
    (the periodic test render included: it is rank 0's own, not a sharded collective).  NEDDF_DIST_BACKEND=gloo selects
    the host backend (tests on one shared GPU)."""
    argv = sys.argv[1:] if argv is None else argv
    cfg = compose(argv)
    cwd = Path.cwd()
    cfg["dataset"]["dataset_dir"] = str(cwd / cfg["dataset"]["dataset_dir"])
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("NEDDF_DIST_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        cfg["trainer"]["device"] = "cuda:%d" % local
    now = datetime.datetime.now()
    run_dir = [str(cwd / "outputs" / now.strftime("%Y-%m-%d") / now.strftime("%H-%M-%S"))]
    if world > 1:
        torch.distributed.broadcast_object_list(run_dir, src=0)          # every rank works in rank 0's directory
    run_dir = Path(run_dir[0])
    if rank == 0:
        (run_dir / ".hydra").mkdir(parents=True)
        yaml.safe_dump(cfg, open(run_dir / ".hydra" / "config.yaml", "w"), sort_keys=False)
        yaml.safe_dump(list(argv), open(run_dir / ".hydra" / "overrides.yaml", "w"))
    if world > 1:
        torch.distributed.barrier()
    os.chdir(run_dir)
    trainer = instantiate(cfg["trainer"], global_config=cfg, _recursive_=False)      # under the common seed (3408)
    trainer.writes_outputs = rank == 0
    if world > 1:
        from neddf_amd.parallel import assert_replicas_identical, sync_parameters
        state = [t for t in trainer.neural_render.state_dict().values() if torch.is_tensor(t) and t.is_floating_point()]
        sync_parameters(state)
        assert_replicas_identical(state)
        if os.environ.get("NEDDF_RUN_PRINT_SIGNATURE"):
            # one write(2) per rank (ranks under a launcher share one pipe: text and newline written separately interleave), and a
            # file per rank beside the run directory for whoever wants it without parsing a shared stream
            sig = "replica_signature[%d]=%.17g" % (rank, float(sum(t.double().sum() for t in state)))
            (run_dir / ("replica_signature.%d" % rank)).write_text(sig + "\n")
            sys.stdout.flush()
            os.write(1, ("\n" + sig + "\n").encode())
        seed_everything(3408 + rank)          # from here on every rank draws its own cameras and pixels
    trainer.run_train()
    if world > 1:
        torch.distributed.destroy_process_group()


def seed_everything(seed: int = 3408) -> None:
    """run.py:30-38"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)


if __name__ == "__main__":
    seed_everything()
    main()
