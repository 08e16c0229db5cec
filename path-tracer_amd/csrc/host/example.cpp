// The reference's demo main (source/example.cpp:9-27) without its window: load a scene,
// construct the Renderer, render.  What a user of the reference writes stays the same; the
// work runs on the GPU through Renderer::render (host/renderer.cpp).
//   pt_example [scene.sdf [out.ppm [width [height [aa_steps [max_ray_bounces [obj_dir]]]]]]]
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>

#include "renderer.hpp"
#include "scene.hpp"

int main(int argc, const char** argv) {
	std::string sdf = argc > 1 ? argv[1] : "../../sdf/cornell.sdf";
	std::string out = argc > 2 ? argv[2] : "../../sdf/img.ppm";
	unsigned img_width = argc > 3 ? (unsigned) std::atoi(argv[3]) : 800;
	unsigned img_height = argc > 4 ? (unsigned) std::atoi(argv[4]) : img_width;
	unsigned aa_steps = argc > 5 ? (unsigned) std::atoi(argv[5]) : 2;
	unsigned bounces = argc > 6 ? (unsigned) std::atoi(argv[6]) : 5;
	try {
		Scene scene = argc > 7 ? load_scene(sdf, argv[7]) : load_scene(sdf);
		Renderer renderer{img_width, img_height, out, aa_steps, bounces};
		std::cout << "shapes " << scene.root->child_count() << "\n";
		std::cout << "lights " << scene.lights.size() << "\n";
		auto start = std::chrono::steady_clock::now();
		renderer.render(scene);
		auto end = std::chrono::steady_clock::now();
		std::chrono::duration<double> elapsed_seconds = end - start;
		std::cout << elapsed_seconds.count() << "s\n";
	} catch (const char* str) {
		std::cout << "Exception: " << str << std::endl;
		return 1;
	} catch (SceneError const& e) {
		std::cout << "Exception: " << e.message << std::endl;
		return 1;
	}
	return 0;
}
