"""Pure-Python helpers for binary glTF 2.0 (test infrastructure): an independent reader to check the product's C++ loader
against, and a writer for synthetic .glb files."""
import json
import struct

import numpy as np

_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def _accessor(doc, binary, index):
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    dt, size = _COMP[acc["componentType"]]
    n = _NCOMP[acc["type"]]
    off = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or size * n
    rows = [np.frombuffer(binary, dtype=dt, count=n, offset=off + k * stride) for k in range(acc["count"])]
    return np.stack(rows) if rows else np.zeros((0, n), dt)


def parse_glb(data):
    """[(positions float32 (V,3), indices uint32 (T,3))] for every TRIANGLES primitive of every mesh, in file order; node
    transforms ignored (as src/util/ModelLoader.cpp:14-41 ignores them)."""
    magic, version, total = struct.unpack_from("<III", data, 0)
    assert magic == 0x46546C67 and version == 2
    at, doc, binary = 12, None, b""
    while at + 8 <= min(total, len(data)):
        length, kind = struct.unpack_from("<II", data, at)
        at += 8
        chunk = data[at:at + length]
        if kind == 0x4E4F534A and doc is None:
            doc = json.loads(chunk.decode("utf-8"))
        elif kind == 0x004E4942 and not binary:
            binary = bytes(chunk)
        at += (length + 3) & ~3
    out = []
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            if prim.get("mode", 4) != 4 or "POSITION" not in prim.get("attributes", {}):
                continue
            pos = _accessor(doc, binary, prim["attributes"]["POSITION"]).astype(np.float32)
            if "indices" in prim:
                idx = _accessor(doc, binary, prim["indices"]).astype(np.uint32).reshape(-1)
            else:
                idx = np.arange(pos.shape[0], dtype=np.uint32)
            idx = idx[: idx.size // 3 * 3].reshape(-1, 3)
            idx = idx[(idx < pos.shape[0]).all(axis=1)]
            if idx.shape[0]:
                out.append((pos, idx))
    return out


def parse_glb_shading(data):
    """[(texcoords float32 (V,2) -- zeros without TEXCOORD_0 --, baseColorFactor rgb, emissiveFactor rgb)] for the primitives
    parse_glb keeps, in the same order; glTF defaults (1,1,1) / (0,0,0) where a primitive or its material says nothing."""
    magic, version, total = struct.unpack_from("<III", data, 0)
    at, doc, binary = 12, None, b""
    while at + 8 <= min(total, len(data)):
        length, kind = struct.unpack_from("<II", data, at)
        at += 8
        chunk = data[at:at + length]
        if kind == 0x4E4F534A and doc is None:
            doc = json.loads(chunk.decode("utf-8"))
        elif kind == 0x004E4942 and not binary:
            binary = bytes(chunk)
        at += (length + 3) & ~3
    out = []
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            if prim.get("mode", 4) != 4 or "POSITION" not in prim.get("attributes", {}):
                continue
            n = doc["accessors"][prim["attributes"]["POSITION"]]["count"]
            uv = np.zeros((n, 2), np.float32)
            if "TEXCOORD_0" in prim["attributes"]:
                a = _accessor(doc, binary, prim["attributes"]["TEXCOORD_0"])
                if a.dtype == np.dtype("<f4") and a.shape == (n, 2):
                    uv = a.astype(np.float32)
            base, emissive = np.ones(3, np.float32), np.zeros(3, np.float32)
            if "material" in prim and prim["material"] < len(doc.get("materials", [])):
                m = doc["materials"][prim["material"]]
                if "baseColorFactor" in m.get("pbrMetallicRoughness", {}):
                    base = np.asarray(m["pbrMetallicRoughness"]["baseColorFactor"][:3], np.float64).astype(np.float32)
                if "emissiveFactor" in m:
                    emissive = np.asarray(m["emissiveFactor"][:3], np.float64).astype(np.float32)
            out.append((uv, base, emissive))
    return out


def write_glb(primitives):
    """primitives: list of dicts(positions (V,3) float32, indices (T,3) or None, index_type 5121/5123/5125, interleave bool,
    mode int).  Returns the bytes of a .glb holding one mesh per primitive."""
    binary = bytearray()
    views, accessors, meshes = [], [], []

    def add_view(raw, stride=None):
        while len(binary) % 4:
            binary.append(0)
        v = {"buffer": 0, "byteOffset": len(binary), "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        binary.extend(raw)
        views.append(v)
        return len(views) - 1

    for prim in primitives:
        pos = np.ascontiguousarray(prim["positions"], np.float32)
        if prim.get("interleave"):
            # position + a dummy normal interleaved: exercises byteStride and accessor.byteOffset
            both = np.concatenate([np.full_like(pos, 7.0), pos], axis=1).astype("<f4")
            view = add_view(both.tobytes(), 24)
            accessors.append({"bufferView": view, "byteOffset": 12, "componentType": 5126, "count": int(pos.shape[0]), "type": "VEC3"})
        else:
            view = add_view(pos.astype("<f4").tobytes())
            accessors.append({"bufferView": view, "componentType": 5126, "count": int(pos.shape[0]), "type": "VEC3"})
        p = {"attributes": {"POSITION": len(accessors) - 1}, "mode": prim.get("mode", 4)}
        if prim.get("indices") is not None:
            ct = prim.get("index_type", 5125)
            idx = np.ascontiguousarray(prim["indices"]).reshape(-1).astype(_COMP[ct][0])
            view = add_view(idx.tobytes())
            accessors.append({"bufferView": view, "componentType": ct, "count": int(idx.size), "type": "SCALAR"})
            p["indices"] = len(accessors) - 1
        meshes.append({"primitives": [p], "name": "m\"%d\\" % len(meshes)})
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(binary)}], "bufferViews": views, "accessors": accessors,
           "meshes": meshes, "nodes": [{"mesh": k, "translation": [1.0, 2.0, 3.0]} for k in range(len(meshes))],
           "scenes": [{"nodes": list(range(len(meshes)))}], "scene": 0, "extras": {"flag": True, "nothing": None, "n": -1.5e-3}}
    text = json.dumps(doc).encode("utf-8")
    text += b" " * (-len(text) % 4)
    while len(binary) % 4:
        binary.append(0)
    total = 12 + 8 + len(text) + 8 + len(binary)
    return struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(text), 0x4E4F534A) + text + \
        struct.pack("<II", len(binary), 0x004E4942) + bytes(binary)
