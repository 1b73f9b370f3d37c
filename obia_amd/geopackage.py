"""Minimal GeoPackage (OGC 12-128r14) writer / reader for the segment tables of this path -- stdlib only.

The reference ends its tiled driver with ``all_segments.to_file(output_dir/"segments.gpkg", driver="GPKG")``
(obia/utils/tiling.py:289-291: columns ``geometry`` and ``segment_id``) and ``Segments.write_segments`` with
``GeoDataFrame.to_file`` (obia/segmentation/segment.py:55-60); both go through GDAL.  A GeoPackage is an SQLite database
with three metadata tables and one feature table whose geometry column holds GeoPackageBinary blobs (an 8-byte header,
an envelope, then standard WKB), so the file can be written with ``sqlite3`` and the WKB that obia_amd.polygons already
produces -- no geo stack on the writing side.  geopandas / QGIS / ogr2ogr read the result.
"""
import os
import sqlite3
import struct

import numpy as np

_APPLICATION_ID = 0x47504B47      # 'GPKG'
_USER_VERSION = 10200             # GeoPackage 1.2

_SRS_ROWS = [
    ("Undefined cartesian SRS", -1, "NONE", -1, "undefined", "undefined cartesian coordinate reference system"),
    ("Undefined geographic SRS", 0, "NONE", 0, "undefined", "undefined geographic coordinate reference system"),
    ("WGS 84 geodetic", 4326, "EPSG", 4326,
     'GEOGCS["WGS 84",DATUM["WGS_1984",SPHEROID["WGS 84",6378137,298.257223563,AUTHORITY["EPSG","7030"]],'
     'AUTHORITY["EPSG","6326"]],PRIMEM["Greenwich",0,AUTHORITY["EPSG","8901"]],UNIT["degree",0.0174532925199433,'
     'AUTHORITY["EPSG","9122"]],AUTHORITY["EPSG","4326"]]', "longitude/latitude coordinates in decimal degrees on the WGS 84 spheroid"),
]


def _wkb_envelope(wkb):
    """(minx, maxx, miny, maxy) of a little-endian WKB Polygon / MultiPolygon (the two types obia_amd.polygons emits)."""
    def poly(off):
        nr, = struct.unpack_from("<I", wkb, off + 5)
        off += 9
        lo = hi = None
        for _ in range(nr):
            n, = struct.unpack_from("<I", wkb, off)
            pts = np.frombuffer(wkb, "<f8", 2 * n, off + 4).reshape(n, 2)
            mn, mx = pts.min(0), pts.max(0)
            lo = mn if lo is None else np.minimum(lo, mn)
            hi = mx if hi is None else np.maximum(hi, mx)
            off += 4 + 16 * n
        return off, lo, hi
    t, = struct.unpack_from("<I", wkb, 1)
    if t == 3:
        _, lo, hi = poly(0)
    elif t == 6:
        npoly, = struct.unpack_from("<I", wkb, 5)
        off, lo, hi = 9, None, None
        for _ in range(npoly):
            off, l2, h2 = poly(off)
            lo = l2 if lo is None else np.minimum(lo, l2)
            hi = h2 if hi is None else np.maximum(hi, h2)
    else:
        raise ValueError(f"unsupported WKB geometry type {t}")
    return float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1])


def gpkg_blob(wkb, srs_id):
    """GeoPackageBinary: 'GP', version 0, flags (little endian, xy envelope), srs_id, envelope, WKB."""
    minx, maxx, miny, maxy = _wkb_envelope(wkb)
    return b"GP" + struct.pack("<BBi4d", 0, 0x03, int(srs_id), minx, maxx, miny, maxy) + bytes(wkb)


def write_geopackage(path, wkb_list, columns, table="segments", srs_epsg=None, geometry_type=None, srs_wkt=None):
    """Write one feature table.  ``wkb_list``: WKB bytes per feature; ``columns``: dict name -> sequence (ints or floats)
    of the same length.  ``srs_epsg``: EPSG code of the coordinates (None: undefined cartesian, srs_id -1).
    ``geometry_type``: the name registered in gpkg_geometry_columns; None = "POLYGON" when every blob is a WKB Polygon,
    "MULTIPOLYGON" when every blob is a MultiPolygon, "GEOMETRY" for a mix (quickshift labels can come out as MultiPolygons:
    strict readers reject a POLYGON table that holds one).  ``srs_wkt``: the WKT definition of a non-4326 EPSG code; without it
    the row says "undefined" and a reader resolves the CRS from organization / organization_coordsys_id (most do)."""
    n = len(wkb_list)
    for k, v in columns.items():
        if len(v) != n:
            raise ValueError(f"column {k} has {len(v)} rows, expected {n}")
    if os.path.exists(path):
        os.remove(path)
    srs_id = -1 if srs_epsg is None else int(srs_epsg)
    con = sqlite3.connect(path)
    try:
        cur = con.cursor()
        cur.execute(f"PRAGMA application_id = {_APPLICATION_ID}")
        cur.execute(f"PRAGMA user_version = {_USER_VERSION}")
        cur.executescript("""
            CREATE TABLE gpkg_spatial_ref_sys (srs_name TEXT NOT NULL, srs_id INTEGER NOT NULL PRIMARY KEY,
                organization TEXT NOT NULL, organization_coordsys_id INTEGER NOT NULL, definition TEXT NOT NULL, description TEXT);
            CREATE TABLE gpkg_contents (table_name TEXT NOT NULL PRIMARY KEY, data_type TEXT NOT NULL, identifier TEXT UNIQUE,
                description TEXT DEFAULT '', last_change DATETIME NOT NULL DEFAULT (strftime('%Y-%m-%dT%H:%M:%fZ','now')),
                min_x DOUBLE, min_y DOUBLE, max_x DOUBLE, max_y DOUBLE, srs_id INTEGER,
                CONSTRAINT fk_gc_r_srs_id FOREIGN KEY (srs_id) REFERENCES gpkg_spatial_ref_sys(srs_id));
            CREATE TABLE gpkg_geometry_columns (table_name TEXT NOT NULL, column_name TEXT NOT NULL, geometry_type_name TEXT NOT NULL,
                srs_id INTEGER NOT NULL, z TINYINT NOT NULL, m TINYINT NOT NULL,
                CONSTRAINT pk_geom_cols PRIMARY KEY (table_name, column_name),
                CONSTRAINT fk_gc_tn FOREIGN KEY (table_name) REFERENCES gpkg_contents(table_name),
                CONSTRAINT fk_gc_srs FOREIGN KEY (srs_id) REFERENCES gpkg_spatial_ref_sys (srs_id));
        """)
        cur.executemany("INSERT INTO gpkg_spatial_ref_sys VALUES (?,?,?,?,?,?)", _SRS_ROWS)
        if srs_id not in (-1, 0, 4326):
            cur.execute("INSERT INTO gpkg_spatial_ref_sys VALUES (?,?,?,?,?,?)",
                        (f"EPSG:{srs_id}", srs_id, "EPSG", srs_id, srs_wkt or "undefined",
                         None if srs_wkt else "definition not carried by the writer: resolve by organization / organization_coordsys_id"))
        if geometry_type is None:
            kinds = {struct.unpack_from("<I", w, 1)[0] & 0xff for w in wkb_list}
            geometry_type = "POLYGON" if kinds <= {3} else ("MULTIPOLYGON" if kinds == {6} else "GEOMETRY")
        names = list(columns)
        types = {}
        for k in names:
            a = np.asarray(columns[k])
            types[k] = "INTEGER" if a.dtype.kind in "iub" else "REAL"
        cols_sql = "".join(f', "{k}" {types[k]}' for k in names)
        cur.execute(f'CREATE TABLE "{table}" (fid INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, geom BLOB{cols_sql})')
        blobs = [gpkg_blob(w, srs_id) for w in wkb_list]
        if n:
            env = np.array([struct.unpack_from("<4d", b, 8) for b in blobs])
            bounds = (float(env[:, 0].min()), float(env[:, 2].min()), float(env[:, 1].max()), float(env[:, 3].max()))
        else:
            bounds = (None, None, None, None)
        pycols = []
        for k in names:
            a = np.asarray(columns[k])
            pycols.append([int(v) for v in a] if types[k] == "INTEGER" else [None if v != v else float(v) for v in a])
        q = f'INSERT INTO "{table}" (geom{"".join(", " + chr(34) + k + chr(34) for k in names)}) VALUES ({",".join("?" * (1 + len(names)))})'
        cur.executemany(q, zip(blobs, *pycols))
        cur.execute("INSERT INTO gpkg_contents (table_name, data_type, identifier, min_x, min_y, max_x, max_y, srs_id) VALUES (?,?,?,?,?,?,?,?)",
                    (table, "features", table, *bounds, srs_id))
        cur.execute("INSERT INTO gpkg_geometry_columns VALUES (?,?,?,?,?,?)", (table, "geom", geometry_type, srs_id, 0, 0))
        con.commit()
    finally:
        con.close()
    return path


def read_geopackage(path, table="segments"):
    """Read a feature table back: (list of WKB bytes, dict column -> list, srs_id).  Checks the container's magic numbers,
    the metadata rows and every blob's GeoPackageBinary header."""
    con = sqlite3.connect(path)
    try:
        cur = con.cursor()
        if cur.execute("PRAGMA application_id").fetchone()[0] != _APPLICATION_ID:
            raise ValueError("not a GeoPackage (application_id)")
        row = cur.execute("SELECT data_type, srs_id FROM gpkg_contents WHERE table_name = ?", (table,)).fetchone()
        if not row or row[0] != "features":
            raise ValueError(f"no feature table {table!r}")
        gcol, srs_id = cur.execute("SELECT column_name, srs_id FROM gpkg_geometry_columns WHERE table_name = ?", (table,)).fetchone()
        info = [r[1] for r in cur.execute(f'PRAGMA table_info("{table}")').fetchall()]
        others = [c for c in info if c not in ("fid", gcol)]
        rows = cur.execute(f'SELECT "{gcol}"{"".join(", " + chr(34) + c + chr(34) for c in others)} FROM "{table}" ORDER BY fid').fetchall()
    finally:
        con.close()
    wkbs, cols = [], {c: [] for c in others}
    for r in rows:
        b = r[0]
        if b[:2] != b"GP":
            raise ValueError("geometry blob without the GeoPackageBinary magic")
        flags = b[3]
        env_bytes = {0: 0, 1: 32, 2: 48, 3: 48, 4: 64}[(flags >> 1) & 7]
        wkbs.append(bytes(b[8 + env_bytes:]))
        for c, v in zip(others, r[1:]):
            cols[c].append(v)
    return wkbs, cols, srs_id


def wkb_rings(wkb):
    """[[ring (n, 2) float64, ...] per polygon part] of a little-endian WKB Polygon / MultiPolygon (exterior ring first)."""
    def poly(off):
        nr, = struct.unpack_from("<I", wkb, off + 5)
        off += 9
        rings = []
        for _ in range(nr):
            n, = struct.unpack_from("<I", wkb, off)
            rings.append(np.frombuffer(wkb, "<f8", 2 * n, off + 4).reshape(n, 2).copy())
            off += 4 + 16 * n
        return off, rings
    t, = struct.unpack_from("<I", wkb, 1)
    if t == 3:
        return [poly(0)[1]]
    parts, off = [], 9
    for _ in range(struct.unpack_from("<I", wkb, 5)[0]):
        off, r = poly(off)
        parts.append(r)
    return parts
