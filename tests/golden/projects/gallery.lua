--[[ A project file written for this repository's loader tests (pyrite_amd/lua_project.py): every construct the
     reference's own test projects use -- locals, nested tables, calls with table / string / parenthesised arguments,
     method calls, field chains, arithmetic on expressions, require -- in one small scene. ]]
local materials = require "materials"

local ball = shape.sphere {radius = 0.8, position = vector(0, 0.8, 0)}
local lamp_ball = ball:with{material = materials.lamp, radius = 0.5, position = ball.position:with{y = 4, z = 1}}

return {
    image = {width = 96, height = 64},

    renderer = renderer.simple {
        pixel_samples = 8,
        spectrum_samples = 6,
        spectrum_bins = 50, -- not a field of the reference's renderer (it is `spectrum_resolution`): ignored, 64 bins
        tile_size = 16,
        bounces = 6,
        light_samples = 2,
    },

    camera = camera.perspective {
        fov = 50,
        transform = transform.look_at {from = vector(0, 2, 8), to = vector(0, 1, 0)},
        focus_distance = 8.0,
        aperture = 0.001,
    },

    world = {
        sky = light_source.d65 * 0.1,
        objects = {
            shape.plane {origin = vector(), normal = vector {y = 1}, material = materials.floor, texture_scale = 4},
            lamp_ball,
            ball:with{material = materials.green, position = ball.position:with{x = -2}},
            ball:with{material = materials.warm, position = ball.position:with{x = 2}},
            ball:with{material = materials.dense_glass, radius = 0.6, position = vector(-0.7, 0.6, 1.5)},
            ball:with{material = materials.rgb_paint, radius = 0.4, position = vector(0.9, 0.4, 2)},
            ball:with{material = materials.glow, radius = 0.3, position = vector(0, 0.3, 3)},
            shape.mesh {
                file = "../textures/color_checker.obj",
                scale = 0.5,
                materials = {color_checker = {surface = material.diffuse {color = texture "../textures/color_checker.png"}}},
            },
            light.point {position = vector(-4, 5, 4), color = light_source.a * 8},
            light.directional {direction = vector(0.3, 0.9, 0.3), width = 0.98, color = light_source.d65 * 0.5},
        },
    },
}
