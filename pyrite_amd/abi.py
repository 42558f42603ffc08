"""ctypes mirror of include/pyrite_gpu.h (the C ABI). Field order and widths must match the header exactly;
tests/test_abi.py checks sizes and that every declared symbol is exported."""
import ctypes as C

PYR_ABI_VERSION = 4

PYR_OK = 0
PYR_ERR_INVALID_ARGUMENT = -1
PYR_ERR_UNSUPPORTED = -2
PYR_ERR_DEVICE = -3
PYR_ERR_OUT_OF_MEMORY = -4

PYR_FLAG_COUNTERS = 1
PYR_FILM_ROWS, PYR_FILM_TILE_BLOCKS = 0, 1
PYR_COMM_ID_BYTES = 128

# PyrOp
OP_NUMBER, OP_VECTOR, OP_RGB, OP_SPECTRUM, OP_COLOR_TEXTURE, OP_MONO_TEXTURE, OP_RGB_SPECTRUM = range(7)
OP_FRESNEL, OP_BLACKBODY, OP_RGB_TO_VECTOR, OP_BINARY, OP_MIX, OP_CLAMP = range(7, 13)
VT_NUMBER, VT_VECTOR, VT_RGB = 0, 1, 2
BIN_ADD, BIN_SUB, BIN_MUL, BIN_DIV = 0, 1, 2, 3
OPERAND_CONSTANT, OPERAND_INPUT, OPERAND_REGISTER = 0, 1, 2
INPUT_WAVELENGTH = 0
INPUT_NORMAL, INPUT_INCIDENT, INPUT_TEXTURE = 0, 1, 2
DEP_WAVELENGTH, DEP_NORMAL, DEP_INCIDENT, DEP_TEXTURE = 0x01, 0x10, 0x20, 0x40
PROGRAM_CONSTANT, PROGRAM_INSTRUCTIONS = 0, 1
OUTPUT_NUMBER, OUTPUT_VECTOR = 0, 1
SPECTRUM_ARRAY, SPECTRUM_CURVE = 0, 1
BSDF_EMISSIVE, BSDF_DIFFUSE, BSDF_MIRROR, BSDF_REFRACTIVE = 0, 1, 2, 3
LAMP_DIRECTIONAL, LAMP_POINT, LAMP_SHAPE = 0, 1, 2
SHAPE_SPHERE, SHAPE_TRIANGLE, SHAPE_PLANE = 0, 1, 2
HIT_NONE = 0xFFFFFFFF

MAX_NUMBER_REGISTERS, MAX_VECTOR_REGISTERS, MAX_RGB_REGISTERS = 16, 8, 8


class PyrGrain(C.Structure):
    _fields_ = [("acc", C.c_float), ("weight", C.c_float)]


class PyrFilmDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bins", C.c_uint32), ("wl_start", C.c_float), ("wl_width", C.c_float)]


class PyrRenderParams(C.Structure):
    _fields_ = [
        ("bounces", C.c_uint32),
        ("pixel_samples", C.c_uint32),
        ("light_samples", C.c_uint32),
        ("spectrum_samples", C.c_uint32),
        ("tile_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("tile_begin", C.c_uint32),
        ("tile_end", C.c_uint32),
        ("film_row_begin", C.c_uint32),
        ("film_row_count", C.c_uint32),
        ("tile_stride", C.c_uint32),
        ("film_layout", C.c_uint32),
    ]


class PyrCamera(C.Structure):
    _fields_ = [("cam_to_world", C.c_float * 16), ("view_plane", C.c_float), ("focus_distance", C.c_float), ("aperture", C.c_float)]


class PyrOperand(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("bits", C.c_uint32)]


class PyrInstr(C.Structure):
    _fields_ = [
        ("op", C.c_uint32),
        ("value_type", C.c_uint32),
        ("operator_", C.c_uint32),
        ("deps", C.c_uint32),
        ("output", C.c_uint32),
        ("a", C.c_uint32),
        ("b", C.c_uint32),
        ("reserved", C.c_uint32),
        ("x", PyrOperand),
        ("y", PyrOperand),
        ("z", PyrOperand),
        ("w", PyrOperand),
    ]


class PyrProgram(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("constant", C.c_float),
        ("first_instr", C.c_uint32),
        ("num_instrs", C.c_uint32),
        ("output_kind", C.c_uint32),
        ("output_reg", C.c_uint32),
        ("num_numbers", C.c_uint32),
        ("num_vectors", C.c_uint32),
        ("num_rgbs", C.c_uint32),
    ]


class PyrSpectrum(C.Structure):
    _fields_ = [("format", C.c_uint32), ("min", C.c_float), ("max", C.c_float), ("offset", C.c_uint32), ("count", C.c_uint32)]


class PyrComponent(C.Structure):
    _fields_ = [
        ("bsdf", C.c_uint32),
        ("color_program", C.c_uint32),
        ("probability_program", C.c_int32),
        ("selection_compensation", C.c_float),
        ("ior", C.c_float),
        ("env_ior", C.c_float),
        ("dispersion", C.c_float),
        ("env_dispersion", C.c_float),
    ]


class PyrMaterial(C.Structure):
    _fields_ = [
        ("first_component", C.c_uint32),
        ("num_components", C.c_uint32),
        ("first_emissive", C.c_uint32),
        ("num_emissive", C.c_uint32),
        ("normal_map_program", C.c_int32),
    ]


class PyrLamp(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("shape_kind", C.c_uint32),
        ("shape_index", C.c_uint32),
        ("color_program", C.c_uint32),
        ("v", C.c_float * 3),
        ("width", C.c_float),
    ]


class PyrTexture(C.Structure):
    _fields_ = [("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("reserved", C.c_uint32), ("offset", C.c_uint64)]


TEXTURE_COLOR, TEXTURE_MONO = 0, 1

_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)


class PyrSceneDesc(C.Structure):
    _fields_ = [
        ("num_triangles", C.c_uint32),
        ("tri_positions", _fp),
        ("tri_normals", _fp),
        ("tri_uvs", _fp),
        ("tri_material", _up),
        ("num_spheres", C.c_uint32),
        ("spheres", _fp),
        ("sphere_tex_scale", _fp),
        ("sphere_material", _up),
        ("num_planes", C.c_uint32),
        ("planes", _fp),
        ("plane_material", _up),
        ("num_lamps", C.c_uint32),
        ("lamps", C.POINTER(PyrLamp)),
        ("num_materials", C.c_uint32),
        ("materials", C.POINTER(PyrMaterial)),
        ("num_components", C.c_uint32),
        ("components", C.POINTER(PyrComponent)),
        ("num_programs", C.c_uint32),
        ("programs", C.POINTER(PyrProgram)),
        ("num_instrs", C.c_uint32),
        ("instrs", C.POINTER(PyrInstr)),
        ("num_spectra", C.c_uint32),
        ("spectra", C.POINTER(PyrSpectrum)),
        ("num_spectrum_floats", C.c_uint32),
        ("spectrum_data", _fp),
        ("rgb_basis", _fp),
        ("rgb_basis_count", C.c_uint32),
        ("rgb_basis_min", C.c_float),
        ("rgb_basis_max", C.c_float),
        ("sky_program", C.c_uint32),
        ("num_textures", C.c_uint32),
        ("textures", C.POINTER(PyrTexture)),
        ("num_texture_floats", C.c_uint64),
        ("texture_data", _fp),
        ("tri_frames", _fp),
        ("plane_frames", _fp),
    ]


class PyrCounters(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("extension_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("triangle_tests", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("plane_tests", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("exposures", C.c_uint64),
    ]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


class PyrHit(C.Structure):
    _fields_ = [("distance", C.c_float), ("shape", C.c_uint32), ("u", C.c_float), ("v", C.c_float)]


class PyrBvhInfo(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_uint32),
        ("num_leaves", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("num_primitives", C.c_uint32),
        ("node_bytes", C.c_uint64),
        ("primitive_bytes", C.c_uint64),
        ("num_wide_nodes", C.c_uint32),
        ("num_pair_records", C.c_uint32),
        ("wide_node_bytes", C.c_uint64),
        ("pair_record_bytes", C.c_uint64),
    ]


class PyrPathInfo(C.Structure):
    _fields_ = [
        ("stage_scheduler", C.c_uint32),
        ("interpreter", C.c_uint32),
        ("scene_in_lds", C.c_uint32),
        ("tape", C.c_uint32),
        ("phase_lanes", C.c_uint32),
        ("reserved", C.c_uint32 * 3),
    ]


class PyrDevelopParams(C.Structure):
    _fields_ = [
        ("step_size", C.c_float),
        ("xyz_scale", C.c_float),
        ("sample_count", C.c_uint32),
        ("filter", _fp),
        ("white_div", _fp),
        ("white_mul", _fp),
        ("xyz_table", _fp),
        ("xyz_count", C.c_uint32),
        ("xyz_min", C.c_float),
        ("xyz_max", C.c_float),
    ]


PyrProgressFn = C.CFUNCTYPE(None, C.c_void_p, C.c_uint8, C.c_char_p)

# Every entry point include/pyrite_gpu.h declares: name -> (restype, argtypes)
ENTRY_POINTS = {
    "pyr_abi_version": (C.c_int, []),
    "pyr_device_count": (C.c_int, []),
    "pyr_last_error": (C.c_char_p, []),
    "pyr_scene_create": (C.c_int, [C.POINTER(PyrSceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "pyr_scene_destroy": (None, [C.c_void_p]),
    "pyr_render_simple": (
        C.c_int,
        [C.c_void_p, C.POINTER(PyrCamera), C.POINTER(PyrFilmDesc), C.POINTER(PyrRenderParams), C.c_void_p, PyrProgressFn, C.c_void_p],
    ),
    "pyr_render_simple_device": (
        C.c_int,
        [C.c_void_p, C.POINTER(PyrCamera), C.POINTER(PyrFilmDesc), C.POINTER(PyrRenderParams), C.c_void_p, C.c_void_p],
    ),
    "pyr_scene_counters": (C.c_int, [C.c_void_p, C.POINTER(PyrCounters)]),
    "pyr_scene_intersect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(PyrCounters)]),
    "pyr_scene_intersect_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pyr_scene_bvh_info": (C.c_int, [C.c_void_p, C.POINTER(PyrBvhInfo)]),
    "pyr_scene_path_info": (C.c_int, [C.c_void_p, C.POINTER(PyrRenderParams), C.POINTER(PyrPathInfo)]),
    "pyr_film_blocks_grains": (C.c_uint64, [C.POINTER(PyrFilmDesc), C.POINTER(PyrRenderParams)]),
    "pyr_film_blocks_assemble_device": (C.c_int, [C.POINTER(PyrFilmDesc), C.POINTER(PyrRenderParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pyr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pyr_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pyr_comm_destroy": (None, [C.c_void_p]),
    "pyr_comm_uses_rccl": (C.c_int, [C.c_void_p]),
    "pyr_comm_status": (C.c_int, [C.c_void_p]),
    "pyr_render_simple_sharded": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(PyrCamera), C.POINTER(PyrFilmDesc), C.POINTER(PyrRenderParams), C.c_void_p, C.c_void_p],
    ),
    "pyr_render_simple_multi": (
        C.c_int,
        [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(PyrCamera), C.POINTER(PyrFilmDesc), C.POINTER(PyrRenderParams), C.c_void_p, PyrProgressFn, C.c_void_p],
    ),
    "pyr_film_develop": (C.c_int, [C.POINTER(PyrFilmDesc), C.c_void_p, C.POINTER(PyrDevelopParams), C.c_void_p, C.c_int]),
    "pyr_film_develop_device": (C.c_int, [C.POINTER(PyrFilmDesc), C.c_void_p, C.POINTER(PyrDevelopParams), C.c_void_p, C.c_int, C.c_void_p]),
}


def bind(lib):
    """Attach restype / argtypes for every entry point; raises AttributeError if a symbol is missing."""
    for name, (restype, argtypes) in ENTRY_POINTS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
