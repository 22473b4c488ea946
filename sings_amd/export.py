"""On-disk splat formats of the reference -- host-side numpy, no GPU work.

* ``save_ply``  : sings/rec/utils/visualize/vis.py:22-61 -- text PLY, one ``vertex`` element with 62 float
  properties in this order: x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity(logit) scale_0..2(log) rot_0..3.
  SH rows are written channel-major (``shs[:, 1:].transpose(1, 2).flatten(1)``: all red coefficients, then green,
  then blue), normals are zeros.  The reference writes through the third-party ``plyfile`` package (absent here):
  header layout and property order follow it; float text is ``%.9g`` (round-trips fp32) -- PARITY UNPINNED for the
  exact digit strings only.
* ``load_ply``  : reader for the files above (text or binary_little_endian float/uchar properties).
* ``ply_to_splat`` : playground/display/convert.py:11-50 -- 32-byte records (xyz f32, exp(scale) f32, rgba u8,
  normalised quaternion u8), sorted by -exp(sum scale) * sigmoid(opacity); vectorised, byte-identical to the
  reference's per-vertex loop (pinned by tests/test_export.py, which runs the reference function on the same file).
* ``checkpoint_keys`` : the keys of ``SinGS.state_dict()`` (sings_hybrid.py:169-199) a consumer of saved models sees.
"""
import os

import numpy as np

SH_C0 = 0.28209479177387814


def construct_list_of_attributes(n_rest=45):
    l = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    l += [f'f_dc_{i}' for i in range(3)]
    l += [f'f_rest_{i}' for i in range(n_rest)]
    l.append('opacity')
    l += [f'scale_{i}' for i in range(3)]
    l += [f'rot_{i}' for i in range(4)]
    return l


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def ply_attributes(human_gs_out, pose='canonical'):
    """[N, 62] float32 attribute matrix exactly as vis.py:41-57 assembles it."""
    xyz = _np(human_gs_out['xyz_canon'] if pose == 'canonical' else human_gs_out['xyz']).astype(np.float32)
    shs = _np(human_gs_out['shs']).astype(np.float32)                       # [N, 16, 3]
    f_dc = np.ascontiguousarray(shs[:, :1].transpose(0, 2, 1)).reshape(len(xyz), -1)
    f_rest = np.ascontiguousarray(shs[:, 1:].transpose(0, 2, 1)).reshape(len(xyz), -1)
    op = _np(human_gs_out['opacity']).astype(np.float32)
    opac = np.log(op / (1 - op))                                             # inverse_sigmoid, utils/general.py
    scale = np.log(_np(human_gs_out['scales_canon']).astype(np.float32))
    rot = _np(human_gs_out['rotq_canon']).astype(np.float32)
    return np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opac.reshape(len(xyz), -1), scale, rot), axis=1)


def save_ply(human_gs_out, path, pose='canonical'):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    attr = ply_attributes(human_gs_out, pose)
    names = construct_list_of_attributes(attr.shape[1] - 17)
    assert len(names) == attr.shape[1]
    with open(path, 'w') as f:
        f.write('ply\nformat ascii 1.0\n')
        f.write(f'element vertex {attr.shape[0]}\n')
        for n in names:
            f.write(f'property float {n}\n')
        f.write('end_header\n')
        np.savetxt(f, attr, fmt='%.9g', newline='\n')


_PLY_TYPES = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1',
              'int': '<i4', 'int32': '<i4', 'uint': '<u4', 'short': '<i2', 'ushort': '<u2', 'char': 'i1'}


def load_ply(path):
    """-> numpy structured array of the single ``vertex`` element (scalar properties only)."""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'{path}: not a PLY file')
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f'{path}: truncated header')
            tok = line.decode('ascii').split()
            if not tok or tok[0] == 'comment':
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError('list properties are not supported')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        dt = np.dtype(props)
        if fmt == 'ascii':
            raw = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2)
            out = np.empty(n, dtype=dt)
            for i, (name, _) in enumerate(props):
                out[name] = raw[:, i]
            return out
        if fmt == 'binary_little_endian':
            return np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n).copy()
        raise ValueError(f'{path}: unsupported PLY format {fmt}')


def ply_to_splat(vert, legacy_promotion=True):
    """``vert``: structured vertex array (load_ply) -> bytes of the .splat file (convert.py:11-50).

    The reference pins numpy 1.23.5 (requirements.txt:3), where a python float times an ``np.float32`` SCALAR is a
    float64: its per-vertex colour is ``0.5 + SH_C0 * float64(f_dc)`` and ``1 / (1 + float64(exp_f32(-opacity)))``
    (``legacy_promotion=True``, default).  Under numpy >= 2 (NEP 50) the same source computes both in float32;
    ``legacy_promotion=False`` reproduces that and is what the golden file generated in this container pins."""
    sc = np.stack([vert['scale_0'], vert['scale_1'], vert['scale_2']], 1)
    # the reference sorts on the file's own dtype (float32 arithmetic in numpy), ascending argsort of the negated key
    key = -np.exp(vert['scale_0'] + vert['scale_1'] + vert['scale_2']) / (1 + np.exp(-vert['opacity']))
    order = np.argsort(key)
    pos = np.stack([vert['x'], vert['y'], vert['z']], 1).astype(np.float32)[order]
    scales = np.exp(sc.astype(np.float32))[order]
    rot = np.stack([vert['rot_0'], vert['rot_1'], vert['rot_2'], vert['rot_3']], 1).astype(np.float32)[order]
    f_dc = np.stack([vert['f_dc_0'], vert['f_dc_1'], vert['f_dc_2']], 1)[order]
    opac = vert['opacity'][order]
    ct = np.float64 if legacy_promotion else np.float32
    e = np.exp(-opac)                                            # float32 in both numpy generations
    color = np.concatenate([ct(0.5) + ct(SH_C0) * f_dc.astype(ct), (ct(1) / (ct(1) + e.astype(ct)))[:, None]], 1)
    rgba = (color * 255).clip(0, 255).astype(np.uint8)
    rq = ((rot / np.linalg.norm(rot, axis=1, keepdims=True)) * 128 + 128).clip(0, 255).astype(np.uint8)
    rec = np.empty(len(order), dtype=np.dtype([('p', '<f4', 3), ('s', '<f4', 3), ('c', 'u1', 4), ('r', 'u1', 4)]))
    rec['p'], rec['s'], rec['c'], rec['r'] = pos, scales, rgba, rq
    return rec.tobytes()


def checkpoint_keys(num_gs_level=1):
    """Keys of SinGS.state_dict() (sings_hybrid.py:169-199)."""
    keys = ['active_sh_degree', 'xyz', 'triplane', 'scaling_multiplier', 'max_radii2D', 'xyz_gradient_accum', 'denom',
            'optimizer', 'spatial_lr_scale', 'betas', 'lbs_weights', 'vertex_label', 'level_id', 'gs_level_mark']
    for i in range(num_gs_level):
        keys += [f'appearance_dec_{i}', f'geometry_dec_{i}']
    return keys
