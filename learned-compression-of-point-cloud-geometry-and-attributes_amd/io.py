"""PLY in / out for voxelised colour point clouds (host-side IO; numpy only).

The reference reads frames with ``open3d.io.read_point_cloud`` (data/utils/RawLoader.py:46,
metrics/metric.py:50) and writes them with ``o3d.t.io.write_point_cloud`` (model/model.py:326-330);
open3d is not available here.  Supported: ``format ascii 1.0`` and ``format binary_little_endian 1.0``
(8iVFB / MVUB / G-PCC output), one ``vertex`` element whose properties include x, y, z and,
optionally, red / green / blue (uchar) — any further properties (normals, alpha, ...) are skipped.
A cloud is a float32 ``[N, 6]`` array: voxel coordinates and rgb in [0, 1] — the layout
``ColorModel.compress`` takes (model/model.py:95-123).
"""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def _parse_header(f):
    if f.readline().strip() != b"ply":
        raise ValueError("not a PLY file")
    fmt, n_vertex, props, in_vertex, other_elements = None, None, [], False, False
    while True:
        line = f.readline()
        if not line:
            raise ValueError("PLY: header is not terminated")
        tok = line.decode("ascii", "replace").split()
        if not tok or tok[0] == "comment" or tok[0] == "obj_info":
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                if other_elements:
                    raise ValueError("PLY: the vertex element must come first")
                n_vertex = int(tok[2])
            else:
                other_elements = True
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError("PLY: list properties on vertices are not supported")
            if tok[1] not in _TYPES:
                raise ValueError("PLY: unknown property type %r" % tok[1])
            props.append((tok[2], _TYPES[tok[1]]))
        elif tok[0] == "end_header":
            break
    if fmt not in ("ascii", "binary_little_endian") or n_vertex is None:
        raise ValueError("PLY: unsupported format %r" % fmt)
    names = [p[0] for p in props]
    if not all(a in names for a in "xyz"):
        raise ValueError("PLY: vertices need x, y, z")
    return fmt, n_vertex, props


def read_ply(path):
    """-> float32 [N, 6] (x, y, z, r, g, b with colours scaled to [0, 1]; zeros when the file has none)."""
    with open(path, "rb") as f:
        fmt, n, props = _parse_header(f)
        names = [p[0] for p in props]
        if fmt == "ascii":
            table = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            if table.shape != (n, len(props)):
                raise ValueError("PLY: expected %d vertices with %d properties" % (n, len(props)))
            col = lambda name: table[:, names.index(name)]
        else:
            dt = np.dtype([(nm, "<" + t) for nm, t in props])
            raw = np.frombuffer(f.read(n * dt.itemsize), dtype=dt)
            if raw.shape[0] != n:
                raise ValueError("PLY: truncated vertex data")
            col = lambda name: raw[name].astype(np.float64)
        out = np.zeros((n, 6), dtype=np.float32)
        for i, a in enumerate("xyz"):
            out[:, i] = col(a)
        if all(c in names for c in ("red", "green", "blue")):
            for i, c in enumerate(("red", "green", "blue")):
                v = col(c)
                is_byte = dict(props)[c] in ("u1", "i1")
                out[:, 3 + i] = v / 255.0 if is_byte else v
    return out


def write_ply(path, cloud, binary=True):
    """cloud: [N, 3] or [N, 6] (rgb in [0, 1]); coordinates are written as float, colours as uchar."""
    c = np.asarray(cloud, dtype=np.float64)
    has_rgb = c.shape[1] >= 6
    n = c.shape[0]
    head = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "element vertex %d" % n,
            "property float x", "property float y", "property float z"]
    if has_rgb:
        head += ["property uchar red", "property uchar green", "property uchar blue"]
    head.append("end_header")
    rgb = np.clip(np.round(c[:, 3:6] * 255.0), 0, 255).astype(np.uint8) if has_rgb else None
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        if binary:
            fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")] + ([("red", "u1"), ("green", "u1"), ("blue", "u1")] if has_rgb else [])
            rec = np.zeros(n, dtype=np.dtype(fields))
            rec["x"], rec["y"], rec["z"] = c[:, 0], c[:, 1], c[:, 2]
            if has_rgb:
                rec["red"], rec["green"], rec["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
            f.write(rec.tobytes())
        else:
            for i in range(n):
                row = "%g %g %g" % (c[i, 0], c[i, 1], c[i, 2])
                if has_rgb:
                    row += " %d %d %d" % (rgb[i, 0], rgb[i, 1], rgb[i, 2])
                f.write((row + "\n").encode("ascii"))
