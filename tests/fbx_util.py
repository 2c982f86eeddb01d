"""Pure-Python helpers for binary FBX 7.x (test infrastructure): an independent reader of Objects/Geometry control points and
polygons, and a writer for small synthetic files (7400 = 32-bit node headers, 7500 = 64-bit; arrays raw or zlib)."""
import struct
import zlib

import numpy as np


def _walk(d, off, wide):
    hdr = "<QQQB" if wide else "<IIIB"
    end, nprops, _plen, nlen = struct.unpack_from(hdr, d, off)
    p = off + struct.calcsize(hdr)
    if end == 0:
        return None, p
    name = d[p:p + nlen].decode("latin1")
    p += nlen
    props = []
    for _ in range(nprops):
        t = chr(d[p])
        p += 1
        if t in "YCIFDL":
            fmt = {"Y": "<h", "C": "<b", "I": "<i", "F": "<f", "D": "<d", "L": "<q"}[t]
            props.append(struct.unpack_from(fmt, d, p)[0])
            p += struct.calcsize(fmt)
        elif t in "fdlib":
            n, enc, clen = struct.unpack_from("<III", d, p)
            p += 12
            raw = d[p:p + clen]
            p += clen
            if enc:
                raw = zlib.decompress(raw)
            props.append(np.frombuffer(raw, dtype={"f": "<f4", "d": "<f8", "l": "<i8", "i": "<i4", "b": "i1"}[t], count=n))
        else:  # S, R
            n = struct.unpack_from("<I", d, p)[0]
            p += 4
            props.append(bytes(d[p:p + n]))
            p += n
    children = []
    null = 25 if wide else 13
    while p + null <= end:
        c, p = _walk(d, p, wide)
        if c is None:
            break
        children.append(c)
    return (name, props, children), end


def parse_fbx(data):
    """[(positions float32 (V,3), indices uint32 (T,3))] per Geometry: one vertex per polygon corner, fan triangulation."""
    assert data[:20] == b"Kaydara FBX Binary  "
    wide = struct.unpack_from("<I", data, 23)[0] >= 7500
    off, top = 27, []
    while off + (25 if wide else 13) <= len(data):
        n, off = _walk(data, off, wide)
        if n is None:
            break
        top.append(n)

    def find(nodes, name):
        for n in nodes:
            if n[0] == name:
                yield n
            yield from find(n[2], name)

    out = []
    for g in find(top, "Geometry"):
        v = next((c[1][0] for c in g[2] if c[0] == "Vertices"), None)
        pvi = next((c[1][0] for c in g[2] if c[0] == "PolygonVertexIndex"), None)
        if v is None or pvi is None:
            continue
        pts = np.asarray(v, np.float64).reshape(-1, 3)
        pos, tris, poly = [], [], []
        for raw in pvi.tolist():
            last = raw < 0
            poly.append(len(pos))
            pos.append(pts[~raw if last else raw].astype(np.float32))
            if last:
                tris += [(poly[0], poly[c - 1], poly[c]) for c in range(2, len(poly))]
                poly = []
        if tris:
            out.append((np.array(pos, np.float32), np.array(tris, np.uint32)))
    return out


def write_fbx(geometries, version=7400, compress=False):
    """geometries: list of (points (N,3) float64, polygons = list of index lists)."""
    wide = version >= 7500

    def prop_array(t, arr):
        raw = np.ascontiguousarray(arr).tobytes()
        enc = 1 if compress else 0
        body = zlib.compress(raw) if compress else raw
        return t.encode() + struct.pack("<III", arr.size, enc, len(body)) + body

    def prop_str(sv):
        return b"S" + struct.pack("<I", len(sv)) + sv

    def node(name, props, children, base):
        """serialises a node that will start at absolute offset `base`"""
        hdr = struct.calcsize("<QQQB" if wide else "<IIIB")
        pbytes = b"".join(props)
        body = bytearray()
        at = base + hdr + len(name) + len(pbytes)
        for c in children:
            blob = c(at)
            body += blob
            at += len(blob)
        if children:
            body += b"\0" * (25 if wide else 13)
            at += 25 if wide else 13
        head = struct.pack("<QQQB" if wide else "<IIIB", at, len(props), len(pbytes), len(name))
        return head + name.encode() + pbytes + bytes(body)

    def geometry(points, polygons, k):
        pvi = []
        for poly in polygons:
            pvi += list(poly[:-1]) + [~poly[-1]]
        kids = [lambda b, a=np.asarray(points, "<f8").reshape(-1): node("Vertices", [prop_array("d", a)], [], b),
                lambda b, a=np.asarray(pvi, "<i4"): node("PolygonVertexIndex", [prop_array("i", a)], [], b),
                lambda b: node("GeometryVersion", [b"I" + struct.pack("<i", 124)], [], b)]
        return lambda b: node("Geometry", [b"L" + struct.pack("<q", 1000 + k), prop_str(b"Geometry::g"), prop_str(b"Mesh")], kids, b)

    header = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", version)
    tops = [lambda b: node("FBXHeaderExtension", [], [lambda bb: node("Creator", [prop_str(b"tests/fbx_util.py")], [], bb)], b),
            lambda b: node("Objects", [], [geometry(p, q, k) for k, (p, q) in enumerate(geometries)], b)]
    out = bytearray(header)
    for t in tops:
        out += t(len(out))
    out += b"\0" * (25 if wide else 13)
    return bytes(out)
