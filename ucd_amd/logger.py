"""Minimal logger with the reference's method names (utils/logger.py:4-92): rank-0 ``info``, optional
scalar sink.  tensorboardX / wandb are out of scope."""
import logging
import os


class Logger:
    def __init__(self, logdir, rank, type="none", debug=False, filename=None, summary=False, step=None):
        self.rank, self.step, self.logdir, self.scalars = rank, step, logdir, []
        self.debug_flag = debug
        if rank == 0:
            logging.basicConfig(level=logging.DEBUG if debug else logging.INFO, format="%(levelname)s:rank0: %(message)s")
            if filename is not None:
                os.makedirs(logdir, exist_ok=True)
                logging.getLogger().addHandler(logging.FileHandler(os.path.join(logdir, filename)))

    def close(self):
        pass

    def add_scalar(self, tag, scalar_value, step=None):
        self.scalars.append((tag if self.step is None else f"{tag}/{self.step}", float(scalar_value), step))

    def add_image(self, *a, **k): pass
    def add_figure(self, *a, **k): pass
    def add_table(self, *a, **k): pass
    def add_results(self, *a, **k): pass

    def print(self, msg):
        logging.info(msg)

    def info(self, msg):
        if self.rank == 0:
            logging.info(msg)

    def debug(self, msg):
        if self.rank == 0 and self.debug_flag:
            logging.info(msg)

    def error(self, msg):
        logging.error(msg)
