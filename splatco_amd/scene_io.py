"""Scene files of a SplatCo-trained model, so that the renderer can load what
the reference's train.py wrote and write what its render.py reads:

  point_cloud/iteration_N/point_cloud.ply   anchor PLY  (scene/gaussian_model.py:640-712)
  point_cloud/iteration_N/checkpoints.pth   the three MLP heads, 'unite' mode (:1015-1090)
  chkpnt<N>.pth                             (feat_planes.state_dict(), contractor.state_dict())
                                            (GaussianModel.capture :368-372, train.py:316, scene/__init__.py:80-94)

The PLY is the format the `plyfile` package writes for PlyData([PlyElement.describe(elements, 'vertex')])
(binary_little_endian 1.0, one `vertex` element, float properties in the order of
construct_list_of_attributes); reading also accepts ascii and big-endian files and any property
order, as plyfile does.  No third-party dependency."""
import os

import numpy as np
import torch
from torch import nn

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_vertices(path):
    """-> structured numpy array of the `vertex` element."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: elements before `vertex` are not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties on `vertex` are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: PLY header without format / vertex element")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2, dtype=np.float64)
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for j, (n, _) in enumerate(props):
                out[n] = rows[:, j]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)


def write_ply_vertices(path, names, columns):
    """columns [N, len(names)] float32 -> binary little-endian PLY, header as plyfile writes it."""
    columns = np.ascontiguousarray(columns, dtype="<f4")
    assert columns.ndim == 2 and columns.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {columns.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(columns.tobytes())


def construct_list_of_attributes(n_offsets, feat_dim, scale_dim=6, rot_dim=4):   # :640-653
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_offset_{i}" for i in range(n_offsets * 3)]
    names += [f"f_anchor_feat_{i}" for i in range(feat_dim)]
    names += ["opacity"] + [f"scale_{i}" for i in range(scale_dim)] + [f"rot_{i}" for i in range(rot_dim)]
    return names


def save_ply(model, path):                                                       # :655-673
    a = model._anchor.detach().cpu().numpy()
    offset = model._offset.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    cols = np.concatenate((a, np.zeros_like(a), offset, model._anchor_feat.detach().cpu().numpy(),
                           model._opacity.detach().cpu().numpy(), model._scaling.detach().cpu().numpy(),
                           model._rotation.detach().cpu().numpy()), axis=1)
    names = construct_list_of_attributes(model._offset.shape[1], model._anchor_feat.shape[1], model._scaling.shape[1],
                                         model._rotation.shape[1])
    write_ply_vertices(path, names, cols)


def load_ply_sparse_gaussian(model, path, device=None):                           # :675-712
    v = read_ply_vertices(path)
    names = v.dtype.names
    dev = device if device is not None else model._anchor.device

    def family(prefix):
        cols = sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
        return np.stack([np.asarray(v[n], dtype=np.float32) for n in cols], axis=1) if cols else np.zeros((len(v), 0), np.float32)

    anchor = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32)
    offsets = family("f_offset")
    offsets = offsets.reshape((offsets.shape[0], 3, -1))
    t = lambda x: torch.tensor(x, dtype=torch.float, device=dev)
    model._anchor_feat = nn.Parameter(t(family("f_anchor_feat")).requires_grad_(True))
    model._offset = nn.Parameter(t(offsets).transpose(1, 2).contiguous().requires_grad_(True))
    model._anchor = nn.Parameter(t(anchor).requires_grad_(True))
    # requires_grad True on all six, as the reference's loader does (:706-712)
    model._opacity = nn.Parameter(t(np.asarray(v["opacity"], dtype=np.float32)[:, None]).requires_grad_(True))
    model._scaling = nn.Parameter(t(family("scale_")).requires_grad_(True))
    model._rotation = nn.Parameter(t(family("rot")).requires_grad_(True))
    return model


def save_mlp_checkpoints(model, path, mode="unite"):                              # :1015-1062
    if mode != "unite":
        raise NotImplementedError("only the reference's default 'unite' checkpoint (checkpoints.pth) is written")
    os.makedirs(path, exist_ok=True)
    ck = {"opacity_mlp": model.mlp_opacity.state_dict(), "cov_mlp": model.mlp_cov.state_dict(),
          "color_mlp": model.mlp_color.state_dict()}
    if getattr(model, "appearance_dim", 0) > 0 and getattr(model, "embedding_appearance", None) is not None:
        ck["appearance"] = model.embedding_appearance.state_dict()            # :1054-1060
    torch.save(ck, os.path.join(path, "checkpoints.pth"))


def load_mlp_checkpoints(model, path, mode="unite"):                              # :1065-1090
    if mode != "unite":
        raise NotImplementedError("'split' checkpoints are TorchScript traces of the CUDA modules")
    ck = torch.load(os.path.join(path, "checkpoints.pth"), map_location="cpu", weights_only=True)
    model.mlp_opacity.load_state_dict(ck["opacity_mlp"])
    model.mlp_cov.load_state_dict(ck["cov_mlp"])
    model.mlp_color.load_state_dict(ck["color_mlp"])
    if getattr(model, "appearance_dim", 0) > 0:                               # :1087-1088
        if getattr(model, "embedding_appearance", None) is None:
            model.set_appearance(ck["appearance"]["embedding.weight"].shape[0])
        model.embedding_appearance.load_state_dict(ck["appearance"])
    return model


def save_scene(model, model_path, iteration):
    """What Scene.save + train.py:316 leave behind for iteration N."""
    pc = os.path.join(model_path, "point_cloud", f"iteration_{iteration}")
    save_ply(model, os.path.join(pc, "point_cloud.ply"))
    save_mlp_checkpoints(model, pc)
    # element 1 = the contractor's buffers (xyz_min / xyz_max, scene/gaussian_model.py:368-372; Scene loads them at
    # scene/__init__.py:93).  The contraction itself is never applied on the render path,
    # so the model only carries the two bounds along.
    torch.save((model.feat_planes.state_dict(), dict(getattr(model, "contractor_state", {}))),
               os.path.join(model_path, f"chkpnt{iteration}.pth"))


def load_scene(model, model_path, iteration, device=None):                        # scene/__init__.py:80-94
    pc = os.path.join(model_path, "point_cloud", f"iteration_{iteration}")
    load_ply_sparse_gaussian(model, os.path.join(pc, "point_cloud.ply"), device)
    load_mlp_checkpoints(model, pc)
    ck = torch.load(os.path.join(model_path, f"chkpnt{iteration}.pth"), map_location="cpu", weights_only=True)
    model.feat_planes.load_state_dict(ck[0], strict=False)
    model.contractor_state = {k: v for k, v in dict(ck[1]).items()} if len(ck) > 1 else {}
    return model
