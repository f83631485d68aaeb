from .prefetch import DevicePrefetcher  # noqa: F401
