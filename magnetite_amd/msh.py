"""Gmsh MSH 4 ASCII reader with the semantics of the reference's parser (mesher.rs:536-704),
so meshes produced by `gmsh geom.geo -2 -o geom.msh` elsewhere (this image has no gmsh) flow into the HIP solver.

What mesher.rs does, kept as is:
  * only the $Nodes and $Elements sections are read ($Entities and everything else skipped, :568-578,675)
  * $Nodes: first line of the section skipped; per entity block "dim tag parametric numNodes", then numNodes tag
    lines, then numNodes coordinate lines "x y z" of which x, y are used (:583-628); node index = tag - 1
  * $Elements: first line skipped; per block "entityDim entityTag elementType numElements", then numElements lines
    "tag n1 n2 n3 ..."; only blocks with entityDim == 2 are kept, nodes = n1..n3 - 1 (:630-673)
  * nodes are placed by index (:677-688); check_ccw runs on every element (:690-693, area < 1.0 quirk)
The reference deletes the .msh afterwards (:701); this reader leaves the file alone.
"""
import numpy as np

from .meshgen import Mesh, check_ccw
from .solver import MagnetiteError


def parse_mesh(mesh_file, apply_check_ccw=True):
    try:
        with open(mesh_file) as f:
            lines = f.read().split("\n")
    except OSError as err:
        raise MagnetiteError("Mesher", f"Unable to open auto-generated mesh file: {err}")
    state, skipped_meta = "limbo", False
    tags, coords, tris = [], [], []
    it = iter(lines)
    for line in it:
        if not line:
            continue
        if line.startswith("$End"):
            state = "limbo"
        if state == "limbo":
            skipped_meta = False
            if line.startswith("$Entities"):
                state = "entities"
            elif line.startswith("$Node"):
                state = "nodes"
            elif line.startswith("$Elements"):
                state = "elements"
            continue
        if state == "entities":
            continue
        if not skipped_meta:
            skipped_meta = True
            continue
        try:
            head = [int(v) for v in line.split(" ")]
        except ValueError:
            raise MagnetiteError("Mesher", f"Unexpected non-int in mesh data {line!r}")
        if state == "nodes":
            n_local = head[3]
            block_tags = [int(next(it)) for _ in range(n_local)]
            for k in range(n_local):
                xyz = [float(c) for c in next(it).split(" ")]
                tags.append(block_tags[k] - 1)
                coords.append((xyz[0], xyz[1]))
        else:
            entity_dim, n_local = head[0], head[3]
            for _ in range(n_local):
                md = [int(v) for v in next(it).strip().split(" ")]
                if entity_dim == 2:
                    tris.append((md[1] - 1, md[2] - 1, md[3] - 1))
    if not tags:
        raise MagnetiteError("Mesher", "mesh file holds no nodes")
    tags = np.asarray(tags, dtype=np.int64)
    xy = np.zeros((len(tags), 2))
    if tags.min() < 0 or tags.max() >= len(tags) or len(np.unique(tags)) != len(tags):
        raise MagnetiteError("Mesher", "node tags are not a permutation of 1..N")  # the reference indexes blindly
    xy[tags] = np.asarray(coords)
    conn = np.asarray(tris, dtype=np.int32).reshape(-1, 3)
    if conn.size and (conn.min() < 0 or conn.max() >= len(tags)):
        raise MagnetiteError("Mesher", "element references a node outside the mesh")
    print(f"info: loaded {len(tags)} nodes and {len(conn)} elements")
    mesh = Mesh(xy, conn, "msh")
    return check_ccw(mesh) if apply_check_ccw else mesh


def write_msh(mesh, path):
    """Minimal MSH 4.1 ASCII file (one surface entity) that parse_mesh -- and gmsh -- read back."""
    N, E = mesh.num_nodes, mesh.num_elements
    with open(path, "w") as f:
        f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
        f.write("$Entities\n0 0 1 0\n1 0 0 0 1 1 0 0 0\n$EndEntities\n")
        f.write(f"$Nodes\n1 {N} 1 {N}\n2 1 0 {N}\n")
        f.write("\n".join(str(i + 1) for i in range(N)) + "\n")
        f.write("\n".join(f"{float(x)!r} {float(y)!r} 0" for x, y in mesh.xy) + "\n$EndNodes\n")
        f.write(f"$Elements\n1 {E} 1 {E}\n2 1 2 {E}\n")
        f.write("\n".join(f"{e + 1} {a + 1} {b + 1} {c + 1}" for e, (a, b, c) in enumerate(mesh.conn)) + "\n$EndElements\n")
