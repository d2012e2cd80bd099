class FromOriginalModelMixin:
    pass


class PeftAdapterMixin:
    pass
