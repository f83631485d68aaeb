from .prefetch import DevicePrefetcher  # noqa: F401
from .trajectory_feed import DeviceTrajectoryFeed  # noqa: F401
