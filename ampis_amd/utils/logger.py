"""detectron2.utils.logger members AMPIS uses: setup_logger() (notebook cell 4), log_every_n_seconds (data_utils.py:28,88)."""
import logging
import sys
import time

_LAST = {}


def setup_logger(output=None, distributed_rank=0, *, color=True, name="ampis_amd", abbrev_name=None):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    if distributed_rank == 0 and not logger.handlers:
        h = logging.StreamHandler(stream=sys.stdout)
        h.setLevel(logging.DEBUG)
        h.setFormatter(logging.Formatter("[%(asctime)s %(name)s]: %(message)s", datefmt="%m/%d %H:%M:%S"))
        logger.addHandler(h)
    return logger


def log_every_n_seconds(lvl, msg, n=1, *, name=None):
    key = (name, msg[:16])
    now = time.time()
    if key not in _LAST or now - _LAST[key] >= n:
        logging.getLogger(name or "ampis_amd").log(lvl, msg)
        _LAST[key] = now
