"""Dump inspect.signature of every callable on the drop-in boundary from the REAL reference into
tests/golden/api_signatures.json (build container only: imports /root/reference through ref_import).

    python tests/golden/make_golden_signatures.py

Each entry: "<reference module>:<qualified name>" -> {"mirror": "<brainfm_amd module>:<qualified name>",
"params": [[name, kind, default-repr-or-null], ...]}.  tests/test_host_cpu.py compares the mirrors' signatures with it:
same parameter names in the same order with the same defaults; a mirror may only ADD parameters that have defaults and
are listed under "allowed_extra" here (each one a recorded, explained deviation)."""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.setup()

# reference callable -> mirror callable
PAIRS = [
    ("Trainer.models:build_model", "brainfm_amd.models:build_model"),
    ("Trainer.models:get_processors", "brainfm_amd.models:get_processors"),
    ("Trainer.models:get_postprocessor", "brainfm_amd.models:get_postprocessor"),
    ("Trainer.models.joiner:MultiInputIndepJoiner.forward", "brainfm_amd.models:MultiInputIndepJoiner.forward"),
    ("Trainer.models.joiner:SegProcessor.forward", "brainfm_amd.models:SegProcessor.forward"),
    ("Trainer.models.joiner:DistProcessor.forward", "brainfm_amd.models:DistProcessor.forward"),
    ("Trainer.models.unet3d.model:UNet3D.get_feature", "brainfm_amd.models:UNet3D.get_feature"),
    ("Trainer.models.head:TaskHead.forward", "brainfm_amd.models:TaskHead.forward"),
    ("utils.test_utils:evaluate_image", "brainfm_amd.test_utils:evaluate_image"),
    ("utils.test_utils:tiling", "brainfm_amd.test_utils:tiling"),
    ("utils.test_utils:zero_crop", "brainfm_amd.test_utils:zero_crop"),
    ("utils.test_utils:center_crop", "brainfm_amd.test_utils:center_crop"),
    ("utils.test_utils:get_deformed_atlas", "brainfm_amd.test_utils:get_deformed_atlas"),
    ("utils.test_utils:prepare_image", "brainfm_amd.test_utils:prepare_image"),
    ("Generator:build_datasets", "brainfm_amd.generator:build_datasets"),
    ("Generator.datasets:BaseGen.__getitem__", "brainfm_amd.generator:BaseGen.__getitem__"),
    ("Generator.datasets:BrainIDGen.__getitem__", "brainfm_amd.generator:BrainIDGen.__getitem__"),
    ("Generator.datasets:BaseGen.generate_deformation", "brainfm_amd.generator:BaseGen.generate_deformation"),
    ("Generator.datasets:BaseGen.deform_grid", "brainfm_amd.generator:BaseGen.deform_grid"),
    ("Generator.datasets:BaseGen.get_contrast", "brainfm_amd.generator:BaseGen.get_contrast"),
    ("Generator.datasets:BaseGen.get_setup_params", "brainfm_amd.generator:BaseGen.get_setup_params"),
    ("Generator.datasets:BaseGen.generate_sample", "brainfm_amd.generator:BaseGen.generate_sample"),
    ("Generator.datasets:BaseGen.augment_sample", "brainfm_amd.generator:BaseGen.augment_sample"),
    ("Generator.datasets:BaseGen.encode_pathology", "brainfm_amd.generator:BaseGen.encode_pathology"),
    ("Generator.datasets:BaseGen.get_pathology_direction", "brainfm_amd.generator:BaseGen.get_pathology_direction"),
    ("Generator.utils:fast_3D_interp_torch", "brainfm_amd.generator_utils:fast_3D_interp_torch"),
    ("Generator.utils:myzoom_torch", "brainfm_amd.generator_utils:myzoom_torch"),
    ("Generator.utils:make_gaussian_kernel", "brainfm_amd.generator_utils:make_gaussian_kernel"),
    ("Generator.utils:gaussian_blur_3d", "brainfm_amd.generator_utils:gaussian_blur_3d"),
    ("Generator.utils:make_affine_matrix", "brainfm_amd.generator_utils:make_affine_matrix"),
    ("Generator.utils:binarize", "brainfm_amd.generator_utils:binarize"),
    ("Generator.utils:resolution_sampler", "brainfm_amd.generator_utils:resolution_sampler"),
    ("Generator.utils:augment_pathology", "brainfm_amd.generator_utils:augment_pathology"),
    ("Generator.utils:add_gamma_transform", "brainfm_amd.generator_utils:add_gamma_transform"),
    ("Generator.utils:add_bias_field", "brainfm_amd.generator_utils:add_bias_field"),
    ("Generator.utils:resample_resolution", "brainfm_amd.generator_utils:resample_resolution"),
    ("Generator.utils:add_noise", "brainfm_amd.generator_utils:add_noise"),
    ("utils.interpol.api:grid_pull", "brainfm_amd.interpol:grid_pull"),
    ("utils.interpol.api:grid_push", "brainfm_amd.interpol:grid_push"),
    ("utils.interpol.api:grid_count", "brainfm_amd.interpol:grid_count"),
    ("utils.interpol.api:grid_grad", "brainfm_amd.interpol:grid_grad"),
    ("utils.interpol.api:spline_coeff_nd", "brainfm_amd.interpol:spline_coeff_nd"),
    ("utils.interpol.resize:resize", "brainfm_amd.interpol:resize"),
    ("ShapeID.perlin3d:generate_perlin_noise_3d", "brainfm_amd.shapeid:generate_perlin_noise_3d"),
    ("ShapeID.perlin3d:generate_shape_3d", "brainfm_amd.shapeid:generate_shape_3d"),
    ("ShapeID.perlin3d:generate_velocity_3d", "brainfm_amd.shapeid:generate_velocity_3d"),
    ("ShapeID.misc:stream_3D", "brainfm_amd.shapeid:stream_3D"),
    ("ShapeID.DiffEqs.pde:AdvDiffPDE.__init__", "brainfm_amd.shapeid:AdvDiffPDE.__init__"),
    ("ShapeID.DiffEqs.pde:AdvDiffPDE.forward", "brainfm_amd.shapeid:AdvDiffPDE.forward"),
    ("ShapeID.DiffEqs.odeint:odeint", "brainfm_amd.shapeid:odeint"),
    ("ShapeID.DiffEqs.adjoint:odeint_adjoint", "brainfm_amd.shapeid:odeint_adjoint"),
    ("utils.misc:MRIread", "brainfm_amd.volio:MRIread"),
    ("utils.misc:MRIwrite", "brainfm_amd.volio:MRIwrite"),
    ("utils.misc:torch_resize", "brainfm_amd.misc:torch_resize"),
    ("utils.misc:myzoom_torch_anisotropic", "brainfm_amd.misc:myzoom_torch_anisotropic"),
    ("utils.misc:align_volume_to_ref", "brainfm_amd.misc:align_volume_to_ref"),
    ("utils.misc:get_ras_axes", "brainfm_amd.misc:get_ras_axes"),
]

# deviations of the mirrors, each explained; everything else must match exactly
ALLOWED_EXTRA = {
    "brainfm_amd.generator:build_datasets": {
        "cases": "the reference globs NIfTI files from a data root inside the dataset (Generator/datasets.py:86-121, "
                 "prepare_paths); the mirror takes the cases as in-memory volumes or file objects -- optional keyword, "
                 "default None keeps the two-argument call valid"},
    "brainfm_amd.shapeid:stream_3D": {
        "multiplier": "generate_velocity_3d multiplies the three components by V_multiplier right after stream_3D "
                      "(ShapeID/perlin3d.py:149-156); the mirror folds that product into the curl kernel -- default 1.0 is "
                      "the reference's stream_3D"},
    "brainfm_amd.shapeid:generate_perlin_noise_3d": {
        "device": "the reference returns a NumPy array and its callers move it to the device (perlin3d.py:144-156); the "
                  "mirror computes it there -- optional keyword"},
    "brainfm_amd.test_utils:get_deformed_atlas": {
        "MNI": "the reference reads the module globals MNI / A that it loads from files/gca.mgz at import "
               "(utils/test_utils.py:38-43); the mirror defaults to the same globals (load_atlas) and lets a caller pass them",
        "A": "see MNI"},
}


def resolve_from_source(mod, qual):
    """For a module that cannot be imported here (utils/test_utils.py reads an atlas file through nibabel at import): the
    signature of a top-level function from the file's syntax tree -- argument list and defaults compiled over an empty
    body, decorators dropped (the reference's are functools.wraps-style and keep the signature)."""
    import ast
    path = os.path.join(ref_import.REF_ROOT, *mod.split(".")) + ".py"
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == qual:
            node.body = [ast.Pass()]
            node.decorator_list = []
            m = ast.Module(body=[node], type_ignores=[])
            ast.fix_missing_locations(m)
            import numpy
            import torch
            ns = {"np": numpy, "torch": torch}
            exec(compile(m, path, "exec"), ns)
            return ns[qual]
    raise AttributeError("%s has no top-level function %s" % (path, qual))


def resolve(spec):
    mod, qual = spec.split(":")
    try:
        obj = __import__(mod, fromlist=["_"])
    except Exception:                                         # noqa: BLE001
        if mod.startswith("brainfm_amd") or "." in qual:
            raise
        return resolve_from_source(mod, qual)
    for part in qual.split("."):
        obj = getattr(obj, part)
    return obj


def describe(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        d = None if p.default is inspect.Parameter.empty else \
            ("<function %s>" % p.default.__name__ if inspect.isfunction(p.default) else repr(p.default))
        out.append([p.name, p.kind.name, d])
    return out


def main():
    table, failed = {}, {}
    for ref, mirror in PAIRS:
        try:
            table[ref] = {"mirror": mirror, "params": describe(resolve(ref))}
        except Exception as e:                                # noqa: BLE001
            failed[ref] = repr(e)
    out = {"signatures": table, "allowed_extra": ALLOWED_EXTRA, "unresolved_in_reference": failed}
    path = os.path.join(HERE, "api_signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s: %d signatures, %d unresolved %s" % (path, len(table), len(failed), failed))


if __name__ == "__main__":
    main()
