// oracle/ref_assets_clang.cpp -- TEST INFRASTRUCTURE. A second, tiny build of the reference's asset loader
// (src/core/assets.cpp, compiled from where it lies) with the ROCm clang++ instead of g++, only to run
// fc::assets::get_image on image FILES: under g++ `nlohmann::json metadata_json{nlohmann::json::parse(file)}`
// (assets.cpp:169) becomes a one-element array and the loader throws (see fcref_load_image in ref_harness.cpp);
// clang++ picks the copy constructor, as MSVC does. Built by oracle/Makefile into _ref/libfcref_assets_clang.so.
#include <cstddef>
#include <cstdio>
#include <exception>
#include <filesystem>
#include <memory>

#include "core/assets.hpp"
#include "core/math.hpp"

extern "C" int fcref_clang_load_image(char const* root, char const* name, int* width, int* height, double* rgb)
{
    std::filesystem::path const back{std::filesystem::current_path()};
    std::filesystem::current_path(root);
    int rc{0};
    {
        fc::assets loader{};
        std::shared_ptr<fc::image> img{};
        try
        {
            img = loader.get_image(name);
        }
        catch(std::exception const& e)
        {
            std::fprintf(stderr, "fcref_clang_load_image: %s\n", e.what());
        }
        if(!img)
            rc = -1;
        else
        {
            fc::vector2i const res{img->get_resolution()};
            *width = res.x;
            *height = res.y;
            if(rgb)
                for(int y{0}; y < res.y; ++y)
                    for(int x{0}; x < res.x; ++x)
                    {
                        fc::vector3 const c{img->rgb({x, y})};
                        double* o{rgb + 3 * (static_cast<std::size_t>(y) * res.x + x)};
                        o[0] = c.x;
                        o[1] = c.y;
                        o[2] = c.z;
                    }
        }
    }
    std::filesystem::current_path(back);
    return rc;
}
