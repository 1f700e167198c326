// model.hpp — WaveFront OBJ loader of the C++ host API (namespace rt::model), mirroring
// src/model.rs of the reference: rt::model::OBJModelBuilder{spec, transform} ->
// rt::model::Model::try_from(...) / rt::model::group_from_obj(...) -> rt::Group.
#pragma once

#include "raytracer.hpp"

#ifdef RT_API_NS
#error "include model.hpp before defining RT_API_NS"
#endif
#define RT_API_NS rt
#include "model.inc"
#undef RT_API_NS
