// CPU check of csrc/fc_assets.hpp: loads a mesh and an image written by ray-tracing_amd/assets.py,
// prints checksums the Python test recomputes, and round-trips the mesh through save_mesh.
#include <cstdio>
#include "../../ray-tracing_amd/csrc/fc_assets.hpp"
int main(int argc, char** argv)
{
    if(argc < 4) return 2;
    const std::string root{argv[1]}, mesh{argv[2]}, image{argv[3]};
    std::string err;
    fcgpu::mesh_data m;
    if(!fcgpu::load_mesh(fcgpu::mesh_path(root, mesh), m, err)) { std::printf("ERR %s\n", err.c_str()); return 1; }
    double ps = 0, ns = 0, us = 0;
    unsigned long long is = 0;
    for(std::size_t i = 0; i < m.positions.size(); ++i) ps += m.positions[i] * double(i % 7 + 1);
    for(std::size_t i = 0; i < m.normals.size(); ++i) ns += m.normals[i] * double(i % 5 + 1);
    for(std::size_t i = 0; i < m.uvs.size(); ++i) us += m.uvs[i] * double(i % 3 + 1);
    for(std::size_t i = 0; i < m.indices.size(); ++i) is += m.indices[i] * (unsigned long long)(i % 11 + 1);
    std::printf("MESH %u %zu %.17g %.17g %.17g %llu\n", m.vertex_count(), m.indices.size(), ps, ns, us, is);
    if(!fcgpu::save_mesh(fcgpu::mesh_path(root, mesh + "_copy"), m, err)) { std::printf("ERR %s\n", err.c_str()); return 1; }
    fcgpu::image_data img;
    if(!fcgpu::load_image(root, image, img, err)) { std::printf("ERR %s\n", err.c_str()); return 1; }
    double sum[3] = {0, 0, 0};
    for(int y = 0; y < img.height; ++y)
        for(int x = 0; x < img.width; ++x)
        {
            double c[3];
            img.rgb(x, y, c);
            for(int k = 0; k < 3; ++k) sum[k] += c[k] * double((x + 2 * y) % 13 + 1);
        }
    std::printf("IMAGE %d %d %.17g %.17g %.17g\n", img.width, img.height, sum[0], sum[1], sum[2]);
    fcgpu::mesh_data bad;
    if(fcgpu::load_mesh(fcgpu::mesh_path(root, "truncated"), bad, err)) { std::printf("ERR truncated mesh accepted\n"); return 1; }
    return 0;
}
