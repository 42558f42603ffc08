-- shared materials of the loader test project (exercises `require`, :with{} on expressions and materials)
local curve = spectrum {format = "curve", points = {{400, 0}, {450, 0.3}, {500, 0}, {550, 1}, {600, 0}}}
local glass = material.refractive {ior = 1.5, color = 1}

return {
    lamp = {surface = material.emissive {color = light_source.d65 * 4}},
    floor = {
        surface = mix(material.mirror {color = 1}, material.diffuse {color = texture("../textures/tiles_color.png")}, fresnel(1.5)),
        normal_map = texture("../textures/tiles_normal.png", "linear") * vector(1, -1, 1),
    },
    green = {surface = material.diffuse {color = curve}},
    warm = {surface = material.diffuse {color = curve:with{points = {{580, 0}, {600, 1}, {610, 1}, {650, 0}}}}},
    dense_glass = {surface = glass:with{ior = 1.7, dispersion = 0.01}},
    rgb_paint = {surface = material.diffuse {color = rgb(0.8, 0.3, 0.1) * 0.9 + 0.05}},
    glow = {surface = material.emissive {color = blackbody(3200) * 2e-13} + material.diffuse {color = 0.5}},
}
