"""obia_amd -- MI355X-native tiled SLIC + per-segment zonal statistics, a drop-in for the hot path of
iosefa/obia (segment(method="slic"), create_tiled_segments).  See DESIGN.md / INTEGRATION.md."""
from .segmentation import slic, quickshift, create_segments, segments_table, segment, Segments, normalize_band  # noqa: F401
from .statistics import zonal_stats, create_objects, stats_columns  # noqa: F401
from .polygons import polygonize, PolygonTable  # noqa: F401
from .geopackage import write_geopackage, read_geopackage  # noqa: F401

__version__ = "0.1.0"
