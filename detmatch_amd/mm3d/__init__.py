"""mmdet3d-side host layer of the DetMatch hot path: box structures, the
OpenPCDet adapter, registries/config, SSL detector + modules, step driver."""


def register_all():
    """Import every module that registers a config-addressable type."""
    from . import losses, openpcdet, runner, ssl, ssl_modules  # noqa: F401
    from .. import mm2d  # noqa: F401
