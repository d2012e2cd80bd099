class CacheMixin:
    pass
