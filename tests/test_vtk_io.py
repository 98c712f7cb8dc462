"""CPU-only checks of the VTK reader/writer plumbing (no GPU): our own files and the WriteVTK.jl default encoding
(appended raw data with vtkZLibDataCompressor) decode to the same arrays."""
import base64
import struct
import zlib

import numpy as np

from waterlily_amd import vtk


def test_reader_decodes_appended_compressed(tmp_path):
    N = (5, 4, 3)
    p = np.arange(np.prod(N), dtype=np.float32).reshape(N, order="F")
    v = np.stack([p, 2 * p, 3 * p])                                    # components first
    blobs, offs, off = [], [], 0
    for a in (v, p):
        raw = np.asfortranarray(a).ravel(order="F").tobytes()
        z = zlib.compress(raw)
        blob = struct.pack("<4Q", 1, len(raw), len(raw), len(z)) + z   # [nblocks, blocksize, lastsize, csize...] data
        offs.append(off)
        blobs.append(blob)
        off += len(blob)
    xml = (f'<?xml version="1.0"?>\n<VTKFile type="ImageData" version="1.0" byte_order="LittleEndian" header_type="UInt64" '
           f'compressor="vtkZLibDataCompressor">\n<ImageData WholeExtent="1 5 1 4 1 3" Origin="0 0 0" Spacing="1 1 1">\n'
           f'<Piece Extent="1 5 1 4 1 3"><PointData>\n'
           f'<DataArray type="Float32" Name="Velocity" NumberOfComponents="3" format="appended" offset="{offs[0]}"/>\n'
           f'<DataArray type="Float32" Name="Pressure" format="appended" offset="{offs[1]}"/>\n'
           f'</PointData></Piece></ImageData>\n<AppendedData encoding="raw">\n_').encode() + b"".join(blobs) + \
        b"\n</AppendedData>\n</VTKFile>\n"
    f = tmp_path / "jl.vti"
    f.write_bytes(xml)
    d = vtk.read_vti(str(f))
    assert np.array_equal(d["Pressure"], p) and np.array_equal(d["Velocity"], v)


def test_b64_inline_roundtrip(tmp_path):
    a = np.linspace(0, 1, 24, dtype=np.float64)
    txt = vtk._b64(a)
    hl = 4 * ((8 + 2) // 3)
    back = vtk._decode(base64.b64decode(txt[:hl]) + base64.b64decode(txt[hl:]), np.float64, False, np.uint64)
    assert np.array_equal(back, a)
