#!/usr/bin/env python3
"""Export the weights of the reference's trained model examples/encrypted_mnist/mnist_conv.bson (a Flux Chain saved with
BSON.jl) to tests/golden/mnist_conv.npz.  BUILD CONTAINER ONLY: reads /root/reference, which does not exist on the GPU box;
the .npz (data: four weight / bias arrays) is what travels.  BSON is the documented MongoDB binary format; BSON.jl stores a
Julia array as {tag: "array", type: <datatype>, size: [...], data: <binary>} (column-major) and shares sub-objects through
{tag: "backref", ref: k} into the top-level `_backrefs` list.

usage: python tools/export_mnist_bson.py [/root/reference/examples/encrypted_mnist/mnist_conv.bson]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_doc(buf, off):
    (size,) = struct.unpack_from("<i", buf, off)
    end, off = off + size - 1, off + 4
    out = {}
    while off < end:
        t = buf[off]; off += 1
        z = buf.index(b"\x00", off)
        key = buf[off:z].decode(); off = z + 1
        if t == 0x01:
            val = struct.unpack_from("<d", buf, off)[0]; off += 8
        elif t == 0x02:
            (n,) = struct.unpack_from("<i", buf, off); val = buf[off + 4:off + 4 + n - 1].decode(); off += 4 + n
        elif t in (0x03, 0x04):
            val, off = parse_doc(buf, off)
            if t == 0x04:
                val = [val[str(i)] for i in range(len(val))]
        elif t == 0x05:
            (n,) = struct.unpack_from("<i", buf, off); val = bytes(buf[off + 5:off + 5 + n]); off += 5 + n
        elif t == 0x08:
            val = bool(buf[off]); off += 1
        elif t == 0x0A:
            val = None
        elif t == 0x10:
            val = struct.unpack_from("<i", buf, off)[0]; off += 4
        elif t == 0x12:
            val = struct.unpack_from("<q", buf, off)[0]; off += 8
        else:
            raise ValueError(f"BSON element type {t:#x} not handled")
        out[key] = val
    return out, end + 1


def type_name(t, refs):
    while isinstance(t, dict) and t.get("tag") == "backref":
        t = refs[t["ref"] - 1]
    return ".".join(t["name"]) if isinstance(t, dict) and "name" in t else str(t)


def arrays(node, refs, out, seen):
    """depth-first, in field order: every Float32 array reachable from `node` (backrefs resolved once)"""
    if isinstance(node, dict):
        if node.get("tag") == "backref":
            k = node["ref"]
            if k not in seen:
                seen.add(k)
                arrays(refs[k - 1], refs, out, seen)
            return
        if node.get("tag") == "array" and isinstance(node.get("data"), (bytes, bytearray)):
            tn = type_name(node["type"], refs)
            if tn.endswith("Float32"):
                out.append(np.frombuffer(node["data"], dtype="<f4").reshape(node["size"], order="F").astype(np.float64))
            return
        for v in node.values():
            arrays(v, refs, out, seen)
    elif isinstance(node, list):
        for v in node:
            arrays(v, refs, out, seen)


def main(path):
    buf = open(path, "rb").read()
    doc, _ = parse_doc(buf, 0)
    refs = doc.get("_backrefs", [])
    found = []
    arrays(doc["model"], refs, found, set())
    by_shape = {a.shape: a for a in found}
    print("Float32 arrays in the model:", [a.shape for a in found])
    out = {"conv_w": by_shape[(7, 7, 1, 4)][:, :, 0, :],       # Conv((7,7), 1=>4, stride 3): weight [7][7][1][4]
           "conv_b": by_shape[(4,)],
           "fq1_w": by_shape[(64, 256)], "fq1_b": by_shape[(64,)],   # Dense(256, 64)
           "fq2_w": by_shape[(10, 64)], "fq2_b": by_shape[(10,)]}    # Dense(64, 10)
    dst = os.path.join(ROOT, "tests", "golden", "mnist_conv.npz")
    np.savez_compressed(dst, **{k: v.astype(np.float32) for k, v in out.items()})
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples/encrypted_mnist/mnist_conv.bson")
