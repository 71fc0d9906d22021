"""Minimal TOML writer for the simulation.toml schema (tables, arrays of tables, inline tables for nested shapes and
materials, numbers, strings, booleans, number arrays) — enough to write the frames of a sequence back out.
Floats are written with repr(), which round-trips an f64 exactly, so a frame file parses to the values it was given."""


def _scalar(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, float):
        if v != v:
            return "nan"
        if v in (float("inf"), float("-inf")):
            return "inf" if v > 0 else "-inf"
        r = repr(v)
        return r if any(c in r for c in ".e") else r + ".0"
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_scalar(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{ " + ", ".join(f"{k} = {_scalar(x)}" for k, x in v.items()) + " }"
    raise TypeError(f"cannot write {type(v).__name__} as TOML")


def _is_table_array(v):
    return isinstance(v, list) and len(v) > 0 and all(isinstance(x, dict) for x in v)


def _table(out, prefix, tab, header):
    if header:
        out.append(header)
    later = []
    for k, v in tab.items():
        if _is_table_array(v) and prefix.count(".") < 1:  # [[scene.objects]] etc.; deeper levels go inline
            later.append((k, v))
        elif isinstance(v, dict) and not prefix:
            later.append((k, v))
        else:
            out.append(f"{k} = {_scalar(v)}")
    out.append("")
    for k, v in later:
        name = f"{prefix}{k}"
        if isinstance(v, dict):
            _table(out, name + ".", v, f"[{name}]")
        else:
            for item in v:
                _table(out, name + ".", item, f"[[{name}]]")


def dumps(doc):
    out = []
    _table(out, "", doc, None)
    return "\n".join(out)
