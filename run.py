"""Drop-in launcher: ``python -m torch.distributed.run --nproc_per_node=N run.py --method UCD ...``
(the reference's README.md:35 command line)."""
from ucd_amd.run import cli

if __name__ == "__main__":
    cli()
