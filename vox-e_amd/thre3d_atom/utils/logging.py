"""package logger: `from thre3d_atom.utils.logging import log`"""
import logging as _logging


def get_logger(name: str = "thre3d_atom") -> _logging.Logger:
    logger = _logging.getLogger(name)
    if not logger.handlers:
        handler = _logging.StreamHandler()
        handler.setFormatter(_logging.Formatter("%(asctime)s %(levelname)s %(name)s: %(message)s"))
        logger.addHandler(handler)
        logger.setLevel(_logging.INFO)
        logger.propagate = False
    return logger


log = get_logger()
