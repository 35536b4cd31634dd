"""`python neddf/scripts/run_eval.py <run_dir> [--epoch N]` -- same command line,
inputs (`<run_dir>/.hydra/config.yaml`, `<run_dir>/models/model_{epoch:05}.pth`)
and outputs (`<run_dir>/eval/*.png`, psnr/ssim printout) as the reference's
neddf/scripts/run_eval.py:10-44.  The frozen config is read with PyYAML and the
`dataset.data_split=test` override applied by hand (hydra is not required)."""
from argparse import ArgumentParser
from pathlib import Path

import yaml

from neddf_amd.config import instantiate


def main(argv=None) -> None:
    parser = ArgumentParser()
    parser.add_argument("output_dir", type=Path, help="directory path where models and render are located")
    parser.add_argument("--epoch", type=int, default=2000, help="epoch number of model")
    args = parser.parse_args(argv)
    output_dir = args.output_dir.resolve()
    conf = output_dir / ".hydra" / "config.yaml"
    assert conf.is_file(), conf
    cfg = yaml.safe_load(open(conf))
    cfg["dataset"]["data_split"] = "test"
    trainer = instantiate(cfg["trainer"], global_config=cfg, _recursive_=False)
    trainer.load_pretrained_model(output_dir / "models/model_{:05}.pth".format(args.epoch))
    save_dir = args.output_dir / "eval"
    save_dir.mkdir(exist_ok=True)
    trainer.render_all(save_dir)


if __name__ == "__main__":
    main()
