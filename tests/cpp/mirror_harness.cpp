// tests/cpp/mirror_harness.cpp — drives the C++ host mirror (include/contrast_renderer.hpp) the way an application drives the
// reference: build Paths, Shape::from_paths / Scene, record a render pass, read the results back. It writes everything it produced to
// a binary file; tests/test_cpp_mirror.py builds the same scene through the Python mirror + the oracle and compares byte for byte.
//   usage: mirror_harness <font.ttf> <out.bin>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "contrast_renderer.hpp"

using namespace contrast_renderer;

static void put(std::ofstream& out, const void* p, size_t n) { out.write(static_cast<const char*>(p), (std::streamsize)n); }
static void put_u64(std::ofstream& out, uint64_t v) { put(out, &v, 8); }

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    try {
        std::ifstream font_file(argv[1], std::ios::binary);
        const std::vector<uint8_t> font_data((std::istreambuf_iterator<char>(font_file)), std::istreambuf_iterator<char>());
        Font font("OpenSans", font_data);

        PathBatch batch;
        // Shape 0: a filled rectangle
        batch.add_shape({}, {Path::from_rect({0.0f, 0.0f}, {0.75f, 0.5f})});
        // Shape 1: closed quadratic blob, stroked with round joins
        {
            Path p;
            p.start = {0.5f, 0.0f};
            p.push_integral_quadratic_curve({0.5f, 0.5f}, {0.0f, 0.5f});
            p.push_integral_quadratic_curve({-0.5f, 0.5f}, {-0.5f, 0.0f});
            p.push_rational_quadratic_curve(0.75f, {-0.5f, -0.5f}, {0.0f, -0.5f});
            p.push_line({0.5f, -0.25f});
            StrokeOptions so;
            so.width = 0.125f, so.offset = 0.25f, so.miter_clip = 2.0f, so.closed = true;
            so.curve_approximation = CurveApproximation::UniformTangentAngle(0.25f);
            so.legalize();
            p.stroke_options = so;
            batch.add_shape({DynamicStrokeOptions::Solid(Join::Round, Cap::Butt, Cap::Butt)}, {p});
        }
        // Shape 2: integral + rational cubic fill
        {
            Path p;
            p.start = {-0.5f, -0.25f};
            p.push_integral_cubic_curve({-0.25f, 0.75f}, {0.25f, 0.75f}, {0.5f, -0.25f});
            const float weights[4] = {1.0f, 1.5f, 1.5f, 1.0f};
            p.push_rational_cubic_curve(weights, {0.25f, -0.75f}, {-0.25f, -0.75f}, {-0.5f, -0.25f});
            batch.add_shape({}, {p});
        }
        // Shape 3: the glyphs of "g8" from the native glyph producer (text::paths_of_text), one Shape holding all contours
        batch.add_shape({}, paths_of_text(font, Layout{1.0f, Orientation::LeftToRight, Alignment::Center, Alignment::Center}, U"g8"));
        // Shape 4: an open dashed stroke
        {
            Path p = Path::from_polygon({{-0.75f, -0.5f}, {-0.25f, 0.5f}, {0.25f, -0.5f}, {0.75f, 0.5f}});
            StrokeOptions so;
            so.width = 0.0625f, so.miter_clip = 4.0f;
            so.curve_approximation = CurveApproximation::UniformlySpacedParameters(4);
            p.stroke_options = so;
            batch.add_shape({DynamicStrokeOptions::Dashed(Join::Miter, {DashInterval{0.5f, 1.0f, Cap::Round, Cap::Out}, DashInterval{2.0f, 2.5f, Cap::Butt, Cap::Square}}, 0.25f)}, {p});
        }

        Renderer renderer(0, Configuration{4, 4, 4, 1});
        Scene scene(renderer, batch);
        std::ofstream out(argv[2], std::ios::binary);
        put_u64(out, scene.n_shapes());
        for (uint32_t s = 0; s < scene.n_shapes(); ++s) {
            const ShapeBuffers b = scene.buffers(s);
            put(out, b.vertex_offsets, sizeof(b.vertex_offsets));
            put(out, b.index_offsets, sizeof(b.index_offsets));
            put(out, b.vertex_bytes.data(), b.vertex_bytes.size());
            put(out, b.index_bytes.data(), b.index_bytes.size());
        }
        // instance data: Shape i at (cx, cy) with scale s, column-major mat4
        const float place[5][3] = {{-0.5f, 0.5f, 0.4f}, {0.5f, 0.5f, 0.4f}, {-0.5f, -0.5f, 0.4f}, {0.5f, -0.5f, 0.6f}, {0.0f, 0.0f, 0.9f}};
        const float rgba[5][4] = {{1.0f, 0.0f, 0.0f, 1.0f}, {0.0f, 0.5f, 1.0f, 0.75f}, {0.0f, 1.0f, 0.0f, 0.5f}, {0.25f, 0.25f, 0.25f, 1.0f}, {1.0f, 0.5f, 0.0f, 1.0f}};
        std::vector<float> transforms, colors;
        for (int i = 0; i < 5; ++i) {
            const float m[16] = {place[i][2], 0, 0, 0, 0, place[i][2], 0, 0, 0, 0, 1, 0, place[i][0], place[i][1], 0, 1};
            transforms.insert(transforms.end(), m, m + 16);
            colors.insert(colors.end(), rgba[i], rgba[i] + 4);
        }
        Frame frame(renderer, 160, 128);
        frame.clear();
        scene.render(frame, transforms, colors); // Stencil + Color of every Shape
        std::vector<uint8_t> image = frame.download();
        put(out, image.data(), image.size());

        // a recorded pass: Shape 0 clips Shapes 3 and 4; Shape 1 is drawn after the UnClip; Shape 2 twice (instancing)
        frame.clear();
        RenderPass pass(renderer, frame);
        for (int i = 0; i < 5; ++i) {
            float m[16], c[4];
            for (int k = 0; k < 16; ++k) m[k] = transforms[16 * i + k];
            for (int k = 0; k < 4; ++k) c[k] = colors[4 * i + k];
            if (i == 0) m[0] = 1.2f, m[5] = 1.0f, m[12] = 0.0f, m[13] = 0.0f; // the clip rectangle covers the centre
            pass.push_instance(m, c);
        }
        const float m5[16] = {0.3f, 0, 0, 0, 0, 0.3f, 0, 0, 0, 0, 1, 0, 0.6f, -0.6f, 0, 1}, c5[4] = {0.0f, 0.0f, 1.0f, 1.0f};
        pass.push_instance(m5, c5);
        // the clip is a Shape OBJECT OF ITS OWN (the same rectangle as Shape 0 of the Scene): what it leaves in the stencil attachment clips the
        // Scene's Shapes rendered behind it, as in the reference (renderer.rs:257-266) — the frame keeps the pass state between the objects
        Shape clip = Shape::from_paths(renderer, {}, {Path::from_rect({0.0f, 0.0f}, {0.75f, 0.5f})});
        clip.render(pass, 0, 1, RenderOperation::Stencil);
        pass.set_clip_depth(1);
        clip.render(pass, 0, 1, RenderOperation::Clip);
        pass.render(scene, 3, 3, 4, RenderOperation::Stencil);
        pass.render(scene, 3, 3, 4, RenderOperation::Color);
        pass.render(scene, 4, 4, 5, RenderOperation::Stencil);
        pass.render(scene, 4, 4, 5, RenderOperation::Color);
        pass.set_clip_depth(0);
        clip.render(pass, 0, 1, RenderOperation::UnClip);
        pass.render(scene, 1, 1, 2, RenderOperation::Stencil);
        pass.render(scene, 1, 1, 2, RenderOperation::Color);
        pass.render(scene, 2, 2, 3, RenderOperation::Stencil);
        pass.render(scene, 2, 2, 3, RenderOperation::Color);
        pass.render(scene, 2, 5, 6, RenderOperation::Stencil);
        pass.render(scene, 2, 5, 6, RenderOperation::Color);
        pass.submit();
        image = frame.download();
        put(out, image.data(), image.size());

        // decals in a 3-D scene (main.rs:162-202): perspective instances, depth tested (LessEqual) and written, back faces culled (main.rs:46-49)
        {
            Configuration config{1, 2, 4, 0};
            config.depth_compare = CRH_COMPARE_LESS_EQUAL;
            config.depth_write_enabled = true;
            config.cull_mode = CRH_CULL_BACK; // as the showcase, main.rs:46
            Renderer renderer3d(0, config);
            Scene scene3d(renderer3d, batch);
            Frame frame3d(renderer3d, 160, 128);
            const Mat4 projection = perspective_projection(1.5707964f, 160.0f / 128.0f, 1.0f, 1000.0f);
            std::vector<float> t3d;
            for (int i = 0; i < 5; ++i) {
                const Mat4 m = matrix_multiplication(projection, translation_matrix(place[i][0] * 2.0f, place[i][1] * 1.5f, 1.5f + 0.75f * (float)((i * 3) % 5)));
                t3d.insert(t3d.end(), m.begin(), m.end());
            }
            frame3d.clear();
            std::vector<float> wall((size_t)160 * 128, 1.0f);
            for (int y = 0; y < 128; ++y)
                for (int x = 0; x < 40; ++x) wall[(size_t)y * 160 + x] = 0.25f; // something close to the eye hides the left quarter
            frame3d.upload_depth(wall);
            scene3d.render(frame3d, t3d, colors);
            const std::vector<uint8_t> image3d = frame3d.download();
            const std::vector<float> depth3d = frame3d.download_depth();
            put(out, t3d.data(), t3d.size() * 4);
            put(out, image3d.data(), image3d.size());
            put(out, depth3d.data(), depth3d.size() * 4);
        }

        // error behaviour: the reference's Err(..) values arrive as exceptions with the same variants
        int errors = 0;
        try {
            pass.set_clip_depth(16);
        } catch (const Error& e) {
            errors += e.status == CRH_ERR_CLIP_STACK_OVERFLOW;
        }
        try {
            Renderer bad(0, Configuration{1, 5, 4, 0});
        } catch (const Error& e) {
            errors += e.status == CRH_ERR_NUMBER_OF_STENCIL_BITS_IS_UNSUPPORTED;
        }
        try {
            Path p = Path::from_rect({0.0f, 0.0f}, {1.0f, 1.0f});
            StrokeOptions so;
            so.dynamic_stroke_options_group = 3;
            p.stroke_options = so;
            Shape::from_paths(renderer, {DynamicStrokeOptions::Solid(Join::Miter, Cap::Butt, Cap::Butt)}, {p});
        } catch (const Error& e) {
            errors += e.status == CRH_ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS;
        }
        Shape single = Shape::from_paths(renderer, {}, {Path::from_rect({3.0f, 4.0f}, {1.0f, 2.0f})});
        const ShapeBuffers sb = single.buffers();
        put_u64(out, (uint64_t)errors);
        put_u64(out, sb.vertex_offsets[7]);
        put(out, sb.vertex_bytes.data(), sb.vertex_bytes.size());
        // TextGeometry and its cursor helpers (text.rs:266-352)
        const TextGeometry g = TextGeometry::make(font, Layout{1.0f, Orientation::LeftToRight, Alignment::Center, Alignment::Center}, U"ab\ncd\nef");
        std::printf("geometry %zu %zu %zu %zu %zu %zu %zu %.6f\n", g.lines.size(), g.line_index_from_char_index(4), g.char_index_from_position(g.lines[1].second[1]),
                    g.advance_char_index_by_line_index(4, -1), g.advance_char_index_by_line_index(4, 1), g.advance_char_index_by_line_index(7, 1),
                    byte_offset_of_char_index("a\xc3\xa9\xe2\x82\xac" "b", 2), g.half_extent.second);
        std::printf("ok %u shapes, %d reference errors reproduced\n", scene.n_shapes(), errors);
        return 0;
    } catch (const Error& e) {
        std::fprintf(stderr, "contrast_renderer error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
}
