import logging as _logging

USE_PEFT_BACKEND = False


class _L:
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)


logging = _L()


def scale_lora_layers(model, weight):
    pass


def unscale_lora_layers(model, weight=None):
    pass


def deprecate(*args, **kwargs):
    pass


def is_scipy_available():
    return False
