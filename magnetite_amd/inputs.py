"""the reference's input JSON (mesher.rs:706-930) -> ModelMetadata + boundary rules.

Reads the reference's `input.json` files verbatim:
  metadata.{part_thickness, material_elasticity, poisson_ratio, characteristic_length_min, characteristic_length_max}
                                                                                         (mesher.rs:769-808)
  boundary_conditions.<name>.region.{x,y}_target_{min,max} (optional, default unbounded, mesher.rs:835-860)
  boundary_conditions.<name>.targets.{ux,uy,fx,fy}         (number or null,              mesher.rs:863-868)
Rules keep the file's order (later rules override earlier ones, mesher.rs:913-927); validation errors carry the
reference's messages as "Input error: ...".
"""
import json

from .meshgen import BoundaryRule, apply_boundary_rules
from .solver import MagnetiteError, ModelMetadata


def load_input_file(path):
    """mesher.rs:713-760: the reference's checks in the reference's order, with its messages verbatim (the
    `boundary_conditions` one says "in metadata section" there too, mesher.rs:738-741; after "Error in input file json: "
    comes the json parser's own text -- the `json` crate's there, Python's here)."""
    try:
        with open(path) as f:
            text = f.read()
    except OSError:
        raise MagnetiteError("Input", f"Unable to open input file {path}")
    try:
        doc = json.loads(text)
    except json.JSONDecodeError as err:
        raise MagnetiteError("Input", f"Error in input file json: {err}")
    has = lambda obj, key: isinstance(obj, dict) and key in obj  # JsonValue::has_key is false on non-objects
    if not has(doc, "metadata"):
        raise MagnetiteError("Input", "Input json missing metadata field")
    if not has(doc, "boundary_conditions"):
        raise MagnetiteError("Input", "Input json missing boundary_conditions field in metadata section")
    for key in ("part_thickness", "material_elasticity", "poisson_ratio"):
        if not has(doc["metadata"], key):
            raise MagnetiteError("Input", f"Input json missing {key} field in metadata section")
    return doc


def parse_input_metadata(doc):
    """mesher.rs:769-808"""
    md = doc.get("metadata", {})
    need = (("material_elasticity", "material elasticity"), ("poisson_ratio", "poisson ratio"),
            ("characteristic_length_min", "minimum characteristic length"),
            ("characteristic_length_max", "maximum characteristic length"))
    for key, what in need:
        if not isinstance(md.get(key), (int, float)) or isinstance(md.get(key), bool):
            raise MagnetiteError("Input", f"Input json missing {what}")
    if not isinstance(md.get("part_thickness"), (int, float)):
        raise MagnetiteError("Input", "Input json missing part thickness")  # the reference unwrap()-panics here
    return ModelMetadata(float(md["material_elasticity"]), float(md["poisson_ratio"]), float(md["part_thickness"]),
                         float(md["characteristic_length_min"]), float(md["characteristic_length_max"]))


def parse_boundary_rules(doc):
    """mesher.rs:822-904: region + targets per named rule, validated."""
    rules = []
    for name, rj in doc.get("boundary_conditions", {}).items():
        if "region" not in rj:
            raise MagnetiteError("Input", f"Boundary rule {name} is missing region field")
        if "targets" not in rj:
            raise MagnetiteError("Input", f"Boundary rule {name} is missing target field")
        kw = {}
        for jkey, field in (("x_target_min", "x_min"), ("x_target_max", "x_max"), ("y_target_min", "y_min"),
                            ("y_target_max", "y_max")):
            if jkey in rj["region"]:
                v = rj["region"][jkey]
                if not isinstance(v, (int, float)) or isinstance(v, bool):
                    raise MagnetiteError("Input", f"Bad value for {jkey} in {name}")
                kw[field] = float(v)
        for t in ("ux", "uy", "fx", "fy"):
            v = rj["targets"].get(t)
            kw[t] = float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else None
        rule = BoundaryRule(name, **kw)
        try:
            rule.validate()
        except ValueError as err:
            raise MagnetiteError("Input", str(err).replace("Input error: ", ""))
        rules.append(rule)
    print(f"info: loaded {len(rules)} boundary rules from input file")
    return rules


def problem_from_input(mesh, path):
    """mesher::run minus the meshing (mesher.rs:939-974): input.json + an existing mesh -> flat problem."""
    doc = load_input_file(path)
    md = parse_input_metadata(doc)
    rules = parse_boundary_rules(doc)
    p = apply_boundary_rules(mesh, rules, youngs_modulus=md.youngs_modulus, poisson_ratio=md.poisson_ratio,
                             part_thickness=md.part_thickness)
    p.meta = dict(metadata=md, rules=[r.name for r in rules])
    return p
