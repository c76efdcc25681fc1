"""What the reference's own Python takes from the modules this repository stands in for (THIS container only: reads
/root/reference; the result, tests/golden/dropin_manifest.json, is data — names, arities, file:line — no source text).

    python tests/golden/make_dropin_manifest.py

An AST walk over the files on the CNC path (examples/train_CNC_*.py, examples/utils.py, examples/utils_bpp_acc.py,
examples/radiance_fields/ngp.py) records, for each of `_gridencoder`, `pack_and_align`, `nerfacc`, `torchac`,
`tinycudann`:
  * every name imported from the module or one of its submodules (`from nerfacc.volrend import rendering`);
  * every attribute taken from a module alias (`_backend.grid_encode_forward`, `tcnn.Encoding`) and, where it is
    called, the number of positional arguments and the keyword names of each call site;
  * for imported classes (OccGridEstimator): the methods called and attributes read on any object in that file whose
    names are methods / buffers of the reference class (parsed from the reference's class body), with call arities.
tests/test_dropins.py then checks that after `cnc_amd.install_dropins()` every name resolves and every recorded
call binds to the stand-in's signature.
"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
ROOTS = ("_gridencoder", "pack_and_align", "nerfacc", "torchac", "tinycudann")
FILES = ["examples/train_CNC_nerf_synthetic.py", "examples/train_CNC_tank_temples.py", "examples/utils.py",
         "examples/utils_bpp_acc.py", "examples/radiance_fields/ngp.py"]
CLASS_SOURCES = {"OccGridEstimator": "nerfacc/estimators/occ_grid.py"}
# functions of those files that are not on the CNC path (proposal-network sampling is out of scope);
# what they take from nerfacc is listed under "skipped" instead of "uses"
OFF_PATH_FUNCTIONS = {"render_image_with_propnet"}


def class_members(path, cls):
    """(method names, attribute names assigned on self / registered as buffers) of a class in the reference."""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    methods, attrs = set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.FunctionDef):
                    methods.add(item.name)
                    for sub in ast.walk(item):
                        if isinstance(sub, ast.Attribute) and isinstance(sub.value, ast.Name) and sub.value.id == "self" \
                                and isinstance(sub.ctx, ast.Store):
                            attrs.add(sub.attr)
                        if isinstance(sub, ast.Call) and isinstance(sub.func, ast.Attribute) and sub.func.attr == "register_buffer" \
                                and sub.args and isinstance(sub.args[0], ast.Constant):
                            attrs.add(sub.args[0].value)
    return methods, attrs


def call_shape(call):
    return {"npos": len(call.args), "kw": [k.arg for k in call.keywords if k.arg is not None],
            "star": any(isinstance(a, ast.Starred) for a in call.args) or any(k.arg is None for k in call.keywords)}


def walk_file(rel):
    src = open(os.path.join(REF, rel)).read()
    tree = ast.parse(src)
    alias = {}          # local name -> dotted module path
    names = {}          # local name -> (module path, attribute)
    uses = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] in ROOTS:
                    alias[a.asname or a.name.split(".")[0]] = a.name if a.asname else a.name.split(".")[0]
                    uses.append({"kind": "import", "module": a.name, "line": node.lineno})
        elif isinstance(node, ast.ImportFrom) and node.module and node.level == 0 and node.module.split(".")[0] in ROOTS:
            for a in node.names:
                names[a.asname or a.name] = (node.module, a.name)
                uses.append({"kind": "from", "module": node.module, "name": a.name, "line": node.lineno})
    classes = {local: mod_attr for local, mod_attr in names.items() if mod_attr[1] in CLASS_SOURCES}
    members = {cls: class_members(CLASS_SOURCES[cls], cls) for _, cls in classes.values()}
    parents = {}
    for node in ast.walk(tree):
        for child in ast.iter_child_nodes(node):
            parents[child] = node
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in alias:
            rec = {"kind": "attr", "module": alias[node.value.id], "name": node.attr, "line": node.lineno}
            par = parents.get(node)
            if isinstance(par, ast.Call) and par.func is node:
                rec["call"] = call_shape(par)
            uses.append(rec)
        elif isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in names:
            mod, name = names[node.func.id]
            uses.append({"kind": "call", "module": mod, "name": name, "line": node.lineno, "call": call_shape(node)})
        elif isinstance(node, ast.Attribute) and not (isinstance(node.value, ast.Name) and node.value.id in alias):
            for local, (mod, cls) in classes.items():
                methods, attrs = members[cls]
                par = parents.get(node)
                if node.attr in methods and isinstance(par, ast.Call) and par.func is node and not node.attr.startswith("__"):
                    # a method of that name called on some object: keep it only for objects that plausibly are
                    # instances (their expression mentions 'estimator' / 'grid'), the check is by name + arity
                    text = ast.unparse(node.value)
                    if "estimator" in text.lower() and "entropy" not in text.lower() and "prop" not in text.lower():
                        uses.append({"kind": "method", "module": mod, "name": f"{cls}.{node.attr}", "line": node.lineno,
                                     "call": call_shape(par)})
                elif node.attr in attrs and isinstance(node.ctx, ast.Load):
                    text = ast.unparse(node.value)
                    if "estimator" in text.lower() and "entropy" not in text.lower() and "prop" not in text.lower():
                        uses.append({"kind": "member", "module": mod, "name": f"{cls}.{node.attr}", "line": node.lineno})
    def enclosing(node):
        while node in parents:
            node = parents[node]
            if isinstance(node, ast.FunctionDef):
                return node.name
        return None
    spans = [(n.lineno, n.end_lineno) for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name in OFF_PATH_FUNCTIONS]
    for u in uses:
        u["file"] = rel
        if any(a <= u["line"] <= b for a, b in spans):
            u["off_path"] = True
    return uses


def main():
    uses = []
    for rel in FILES:
        uses += walk_file(rel)
    uses.sort(key=lambda u: (u["module"], u.get("name", ""), u["file"], u["line"]))
    out = {"roots": list(ROOTS), "files": FILES, "off_path_functions": sorted(OFF_PATH_FUNCTIONS),
           "uses": [u for u in uses if not u.get("off_path")], "skipped": [u for u in uses if u.get("off_path")]}
    path = os.path.join(HERE, "dropin_manifest.json")
    json.dump(out, open(path, "w"), indent=0)
    per = {}
    for u in uses:
        per.setdefault(u["module"], set()).add(u.get("name", "<module>"))
    for m in sorted(per):
        print(m, sorted(per[m]))
    print(len(uses), "uses ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
