"""Scene assembly (reference: `src/scene.py`): camera + camera quad + Cornell box
(+ meshes) -> BVH -> flattened records, light lists.  Same entry points and `Scene`
attributes as the reference; the `metalcompute` device buffers it held are plain host
numpy arrays here -- the renderer uploads them through the C-ABI (`include/clive2_amd.h`).

Extensions (not in the reference): a file_spec may carry `"mesh": (vertices, faces)`
instead of `"file_path"` (procedural stand-ins), and `materials=` overrides the
material table (config 3 needs alpha > 0).
"""
import time
import numpy as np

from .camera import Camera
from .bvh import construct_BVH, np_flatten_bvh, FastTreeBox
from .load import (triangles_for_box, fast_load, fast_load_ply, fast_load_obj, get_materials,
                   camera_geometry, surface_area)
from .constants import UNIT_Z, ZERO_VECTOR
from . import struct_types


class Scene:
    def __init__(self, device, pixel_width, pixel_height, camera, triangles, boxes, materials,
                 light_triangles, light_counts, light_surface_areas, light_triangle_indices,
                 camera_triangle_indices):
        self.device = device                      # None: the renderer owns the GPU context
        self.pixel_width, self.pixel_height = pixel_width, pixel_height
        self.camera, self.triangles, self.boxes, self.materials = camera, triangles, boxes, materials
        self.light_triangles, self.light_counts = light_triangles, light_counts
        self.light_surface_areas = light_surface_areas
        self.light_triangle_indices = light_triangle_indices
        self.camera_triangle_indices = camera_triangle_indices

    def with_resolution(self, pixel_width, pixel_height):
        """The same scene at another frame size of the SAME aspect ratio: the camera quad is part of the
        geometry (load.py:261-271) and its size depends on the aspect ratio only, so nothing but the two
        pixel counts of the camera record changes (create_scene at the new size gives the same bytes
        otherwise) and the BVH need not be rebuilt."""
        if pixel_width * self.pixel_height != pixel_height * self.pixel_width:
            raise ValueError("with_resolution keeps the aspect ratio (the camera quad would change otherwise)")
        cam = np.array(self.camera, copy=True)
        cam["pixel_width"], cam["pixel_height"] = pixel_width, pixel_height
        return Scene(self.device, pixel_width, pixel_height, cam, self.triangles, self.boxes, self.materials,
                     self.light_triangles, self.light_counts, self.light_surface_areas,
                     self.light_triangle_indices, self.camera_triangle_indices)

    def validate(self):
        """Host-side shape checks run before every upload (a bad index here would be an
        out-of-bounds access on the GPU)."""
        nb, nt, nm = len(self.boxes), len(self.triangles), len(self.materials)
        b = self.boxes
        inner = b["right"] == 0
        if nb < 1 or nt < 1:
            raise ValueError("scene needs at least one box and one triangle")
        if np.any(b["left"][inner] < 1) or np.any(b["left"][inner] + 1 >= nb):
            raise ValueError("inner box child index out of range")
        leaf = ~inner
        if np.any(b["left"][leaf] < 0) or np.any(b["right"][leaf] > nt) or \
                np.any(b["left"][leaf] >= b["right"][leaf]):
            raise ValueError("leaf triangle range out of range")
        if np.any(self.triangles["material"] < 0) or np.any(self.triangles["material"] >= nm):
            raise ValueError("triangle material index out of range")
        if len(self.light_triangles) < 1:
            raise ValueError("scene has no emitter triangles")
        li = np.asarray(self.light_triangle_indices)
        if np.any(li < 0) or np.any(li >= nt):
            raise ValueError("light triangle index out of range")
        return True


def create_scene(pixel_width=1280, pixel_height=720, cam_center=ZERO_VECTOR, cam_direction=UNIT_Z,
                 file_specs=None, materials=None, verbose=False, bvh_builder="auto", room=None, max_members=None):
    """scene.py:21-104.  Extensions: `materials` (table override), `"mesh": (vertices, faces)` file specs,
    `bvh_builder`, and `room` -- the enclosure as a list of load.Triangle objects in place of the
    Cornell box of load.triangles_for_box() (e.g. a subset of it: an open scene; it must keep an emitter); `max_members` -- the
    builder's leaf size (None = the reference's constant 8, constants.py:28; bvh.construct_BVH)."""
    camera = Camera(center=cam_center, direction=cam_direction, pixel_width=pixel_width,
                    pixel_height=pixel_height, phys_width=pixel_width / pixel_height, phys_height=1)
    soups = [FastTreeBox.from_triangle_objects(camera_geometry(camera) + (triangles_for_box() if room is None else list(room)))]
    for spec in file_specs or ():
        kw = dict(material=spec.get("material", 0), scale=spec.get("scale", 1.0),
                  offset=spec.get("offset", ZERO_VECTOR))
        if "mesh" in spec:
            v, f = spec["mesh"]
            soups.append(fast_load(np.asarray(v) * kw["scale"] + kw["offset"], np.asarray(f), material=kw["material"]))
        elif spec["file_path"].endswith(".ply"):
            soups.append(fast_load_ply(ply_path=spec["file_path"], **kw))
        elif spec["file_path"].endswith(".obj"):
            soups.append(fast_load_obj(obj_path=spec["file_path"], **kw))
        else:
            raise NotImplementedError(spec["file_path"])
    soup = FastTreeBox.concat(soups) if len(soups) > 1 else soups[0]

    t0 = time.time()
    boxes, tris = np_flatten_bvh(construct_BVH(soup, builder=bvh_builder, max_members=max_members))
    if verbose:
        print(f"BVH construction took {time.time() - t0:.4f} seconds")

    light_ids = np.flatnonzero(tris["is_light"]).astype(np.int32)
    cam_ids = np.flatnonzero(tris["is_camera"]).astype(np.int32)
    light_tris = tris[light_ids]
    e1 = (light_tris["v1"] - light_tris["v0"])[:, :3]
    e2 = (light_tris["v2"] - light_tris["v0"])[:, :3]
    areas = (np.linalg.norm(np.cross(e1, e2), axis=1) / 2).astype(np.float32)

    scene = Scene(
        device=None, pixel_width=pixel_width, pixel_height=pixel_height,
        camera=np.array([camera.to_struct()]), triangles=tris, boxes=boxes,
        materials=get_materials() if materials is None else np.asarray(materials, dtype=struct_types.Material),
        light_triangles=light_tris, light_counts=np.array(len(light_ids), dtype=np.int32),
        light_surface_areas=areas, light_triangle_indices=light_ids,
        camera_triangle_indices=cam_ids)
    scene.validate()
    return scene


scene_presets = {
    "empty": dict(cam_center=np.array([0, 1.5, 6]), cam_direction=np.array([0, 0, -1])),
    "teapots": dict(cam_center=np.array([7, 0, 8]), cam_direction=np.array([-1, 0, -1]), file_specs=[
        dict(file_path="../resources/teapot.obj", offset=np.array([0, 0, 2.5]), material=5),
        dict(file_path="../resources/teapot.obj", offset=np.array([0, 0, -2.5]), material=0)]),
}
for _name, _file in (("dragon", "dragon_vrip_res3.ply"), ("medium-dragon", "dragon_vrip_res2.ply"),
                     ("big-dragon", "dragon_vrip.ply")):
    scene_presets[_name] = dict(
        cam_center=np.array([0, 1.5, 7.5]), cam_direction=np.array([0, 0, -1]),
        file_specs=[dict(file_path="../resources/" + _file, offset=np.array([0, -4, 0]),
                         material=5, scale=50)])


def _preset(name):
    if name not in scene_presets:
        raise ValueError(f"Preset '{name}' not found.")
    return scene_presets[name]


def create_scene_from_preset(preset_name, pixel_width=1280, pixel_height=720, **kw):
    p = _preset(preset_name)
    return create_scene(pixel_width=pixel_width, pixel_height=pixel_height, cam_center=p["cam_center"],
                        cam_direction=p["cam_direction"], file_specs=p.get("file_specs"), **kw)


def create_scene_from_preset_with_params(preset_name, pixel_width=1280, pixel_height=720,
                                         frame_idx=0, total_frames=1, **kw):
    """Turntable camera on a radius-7.5 circle at height 1.5 (scene.py:223-245)."""
    p = _preset(preset_name)
    theta = 2 * np.pi * frame_idx / total_frames
    s, c = np.sin(theta), np.cos(theta)
    return create_scene(pixel_width=pixel_width, pixel_height=pixel_height,
                        cam_center=np.array([s * 7.5, 1.5, c * 7.5]),
                        cam_direction=np.array([-s, 0, -c]), file_specs=p.get("file_specs"), **kw)
