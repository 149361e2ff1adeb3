"""Reads the reference's material test matrix out of its own test source and writes it down as data.

    python tests/golden/make_material_matrix.py            # needs /root/reference; writes tests/golden/material_matrix.json

/root/reference/tests/test_shading.cpp:359-1919 holds ninety `test_<name>` functions (tests/test_aux_channels.cpp a ninety-first of its own shape), every one of them a material descriptor
(shading_node_desc_t or principled_mat_desc_t, field by field), a texture list, a scene variant (eTestScene: which lights, which camera
extras), a sample count and a denoise / region / adaptive-sampling switch, handed to run_material_test.  The golden images those tests
compare with cannot be reproduced here (env.bin and most textures are absent from the checkout), but the MATRIX is the
reference's own statement of what a backend has to get right -- so it is extracted mechanically (no value is typed by hand) and
tests/test_material_matrix.py renders every entry with the HIP backend and with the live oracle on the reference's own test meshes.

This script only parses text; nothing of the reference is compiled or copied: the output holds names, numbers and line references."""
import json
import os
import re
import sys

SRC = "/root/reference/tests/test_shading.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "material_matrix.json")

PSNR_NAMES = {"DefaultMinPSNR": 30.0, "FastMinPSNR": 28.0, "VeryFastMinPSNR": 25.0}  # test_shading.cpp:351-353


def number(text, consts):
    text = text.strip()
    if text in consts:
        return consts[text]
    if text in PSNR_NAMES:
        return PSNR_NAMES[text]
    if text in ("true", "false"):
        return text == "true"
    m = re.fullmatch(r"(-?[0-9.]+(?:e-?[0-9]+)?)f?", text)
    if not m:
        raise ValueError(f"cannot read the value {text!r}")
    v = float(m.group(1))
    return int(v) if re.fullmatch(r"-?[0-9]+", m.group(1)) else v


def value(text, consts):
    text = text.strip()
    m = re.fullmatch(r"(?:Ray::)?TextureHandle\{(\d+)\}", text)
    if m:
        return {"texture": int(m.group(1))}
    m = re.fullmatch(r"Ray::eShadingNode::(\w+)", text)
    if m:
        return m.group(1)
    return number(text, consts)


def split_args(text):
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
            continue
        depth += ch in "({"
        depth -= ch in ")}"
        cur += ch
    out.append(cur.strip())
    return out


def parse(src):
    text = open(src).read()
    lines = text.split("\n")
    starts = [(i, re.match(r"void test_(\w+)\(const char \*arch_list\[\]", ln).group(1)) for i, ln in enumerate(lines)
              if re.match(r"void test_(\w+)\(const char \*arch_list\[\]", ln)]
    tests = []
    for first, fn in starts:
        last = next(j for j in range(first + 1, len(lines)) if lines[j] == "}")
        body = "\n".join(lines[first:last])
        consts = {m.group(2): number(m.group(3), {}) for m in re.finditer(r"const (int|double|float) (\w+) = ([^;]+);", body)}
        decls = {m.group(2): m.group(1) for m in re.finditer(r"Ray::(shading_node_desc_t|principled_mat_desc_t) (\w+);", body)}
        call = re.search(r"run_material_test\((.*?)\);", body, re.S)
        args = split_args(" ".join(call.group(1).split()))
        assert args[0] == "arch_list" and args[1] == "preferred_device", args
        name = args[2].strip('"')
        var = args[3]
        rest = args[4:]
        long_form = len(rest) > 2 and (rest[2] == "VarianceThreshold" or re.fullmatch(r"[0-9.]+f", rest[2]) is not None)
        if long_form:  # (min_samples, max_samples, variance_threshold, min_psnr, pix_thres, denoise, partial, caching, textures, scene)
            keys = ["min_samples", "max_samples", "variance_threshold", "min_psnr", "pix_thres", "denoise", "partial", "caching", "textures", "scene"]
        else:          # (samples, min_psnr, pix_thres, denoise, partial, textures, scene)
            keys = ["samples", "min_psnr", "pix_thres", "denoise", "partial", "textures", "scene"]
        given = dict(zip(keys, rest))
        entry = {"name": name, "function": "test_" + fn, "line": first + 1, "desc": decls[var]}
        if long_form:
            entry["min_samples"], entry["max_samples"] = number(given["min_samples"], consts), number(given["max_samples"], consts)
            entry["variance_threshold"] = number(given["variance_threshold"], consts)
        else:
            entry["min_samples"] = entry["max_samples"] = number(given["samples"], consts)
            entry["variance_threshold"] = 0.0
        entry["min_psnr"], entry["pix_thres"] = number(given["min_psnr"], consts), number(given["pix_thres"], consts)
        entry["denoise"] = given.get("denoise", "eDenoiseMethod::None").split("::")[-1]
        entry["partial"] = given.get("partial", "false") == "true"
        entry["caching"] = given.get("caching", "false") == "true"
        entry["scene"] = given.get("scene", "eTestScene::Standard").split("::")[-1]
        fields = {}
        for m in re.finditer(r"^\s*" + re.escape(var) + r"\.(\w+)(?:\[(\d)\])? = ([^;]+);", body, re.M):
            field, index, v = m.group(1), m.group(2), value(m.group(3), consts)
            if index is None:
                fields[field] = v
            else:
                fields.setdefault(field, [None, None, None])[int(index)] = v
        entry["fields"] = fields
        tex = re.search(r"const char \*textures\[\] = \{(.*?)\};", body, re.S)
        entry["textures"] = [os.path.basename(t) for t in re.findall(r'"([^"]+)"', tex.group(1))] if (tex and given.get("textures") == "textures") else []
        tests.append(entry)
    return tests


def parse_aux(src):
    """tests/test_aux_channels.cpp:20-72: one more material (five textures, alpha among them) on the Standard scene, 14 samples, judged on the base
    colour, normals and depth images instead of the beauty frame"""
    body = open(src).read()
    consts = {m.group(1): number(m.group(2), {}) for m in re.finditer(r"\b(SampleCount|\w+_MinPSNR) = ([0-9.]+)", body)}
    fields = {}
    for m in re.finditer(r"^\s*mat_desc\.(\w+)(?:\[(\d)\])? = ([^;]+);", body, re.M):
        fields[m.group(1)] = value(m.group(3), consts)
    tex = re.search(r"const char \*textures\[\] = \{(.*?)\};", body, re.S)
    call = re.search(r"setup_test_scene\(threads, \*scene, (-?\d+), ([0-9.]+)f, mat_desc, textures, eTestScene::(\w+)\)", body)
    line = body[:body.index("void test_aux_channels")].count("\n") + 1
    return {"name": "aux_channels", "function": "test_aux_channels", "line": line, "source": "tests/test_aux_channels.cpp", "desc": "principled_mat_desc_t",
            "min_samples": int(call.group(1)), "max_samples": consts["SampleCount"], "variance_threshold": float(call.group(2)),
            "min_psnr": min(consts["BaseColor_MinPSNR"], consts["Normals_MinPSNR"], consts["Depth_MinPSNR"]), "pix_thres": 0, "denoise": "None",
            "partial": False, "caching": False, "scene": call.group(3), "fields": fields,
            "textures": [os.path.basename(t) for t in re.findall(r'"([^"]+)"', tex.group(1))]}


def parse_all(src=SRC):
    return parse(src) + [parse_aux(os.path.join(os.path.dirname(src), "test_aux_channels.cpp"))]


def main():
    if not os.path.exists(SRC):
        sys.exit("needs /root/reference (this container): the matrix is committed as tests/golden/material_matrix.json")
    tests = parse_all(SRC)
    with open(OUT, "w") as f:
        json.dump({"source": "tests/test_shading.cpp + tests/test_aux_channels.cpp of the reference, read by tests/golden/make_material_matrix.py", "tests": tests}, f, indent=1)
    kinds = {}
    for t in tests:
        kinds[t["scene"]] = kinds.get(t["scene"], 0) + 1
    print(f"{len(tests)} tests -> {OUT}")
    print("scene variants:", kinds)
    print("textures:", sorted({x for t in tests for x in t["textures"]}))


if __name__ == "__main__":
    main()
