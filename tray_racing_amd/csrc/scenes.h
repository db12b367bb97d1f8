// scenes.h — procedural stand-in scenes and model loading (see scenes.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace trx {

bool gen_scene(const std::string &name, uint64_t n_tris, uint64_t seed, std::vector<float> &verts,
               std::vector<uint64_t> &objects);
bool scene_camera(const std::string &name, float eye[3], float look_at[3], float *fov_deg);
bool load_model(const std::string &path, std::vector<float> &verts, std::vector<uint64_t> &objects);
// A scene file of the reference (assets/scenes/*.ron: model_path, camera(eye, look_at, fov)): the model's path resolved by
// the reference's rule (src/main.rs:271-284) and the camera.  false = unreadable or not such a file.
bool parse_scene_ron(const std::string &path, std::string &model_path, float eye[3], float look_at[3], float *fov_deg);

} // namespace trx
