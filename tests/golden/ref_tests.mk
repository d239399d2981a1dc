# tests/golden/ref_tests.mk -- TEST INFRASTRUCTURE.
#
# The reference's OWN test programs, compiled UNMODIFIED from the sources where they lie under $(REF) and linked against THIS repository's library
# (embree_amd/lib/libembree4_mi355.so) instead of the reference's:
#     tests/golden/_bin/ref_verify              tutorials/verify/verify.cpp   (embree_verify: tutorials/verify/CMakeLists.txt:8-13)
#     tests/golden/_bin/ref_triangle_geometry   tutorials/triangle_geometry   (configs[0] of BASELINE.json; tutorials/triangle_geometry/CMakeLists.txt, GLFW off)
# No file of the reference is copied; objects and binaries go to tests/golden/_bin/ (git-ignored, travels to the GPU box with gpurun like the built library).
# The reference's CMake is not run: the source lists below restate tutorials/common/{scenegraph,image,lights,tutorial}/CMakeLists.txt; the configured headers
# (config.h, rtcore_config.h) and the objects of its `sys math simd lexers tasking` libraries are the ones oracle/ref.mk produced (oracle/_ref/gen, oracle/_ref/obj/common).
#
# verify.cpp includes "../../include/embree4/rtcore.h" by relative path, i.e. the reference's own header: tests/test_abi.py holds this repository's header to the same
# prototypes, enums and struct layouts, so the binary is what a user of the reference gets when the shared library underneath is swapped.
# Two of its tests (GeometryStateTest, SceneCheckModifiedGeometryTest, verify.cpp:4480-4600) reach into the reference's internal classes (kernels/common/geometry.cpp,
# scene_verify.cpp are part of embree_verify for them); those are not triangle-path tests, are never selected with --run here, and their internal symbols stay
# unresolved (-Wl,--unresolved-symbols=ignore-all): calling them would crash, not calling them costs nothing.
#
#   make -f tests/golden/ref_tests.mk -j8
REF ?= /root/reference
OUT := tests/golden/_bin
OBJ := $(OUT)/obj
GEN := oracle/_ref/gen
LIBDIR := embree_amd/lib
CXX ?= g++
CXXFLAGS := -std=c++11 -O2 -DNDEBUG -fPIC -fsigned-char -flax-vector-conversions -fno-strict-aliasing -w \
            -mf16c -mavx2 -mfma -mlzcnt -mbmi -mbmi2 -DTASKING_INTERNAL -DEMBREE_TARGET_AVX2 \
            -I$(GEN)/kernels/common -I$(GEN)/kernels/bvh -I$(GEN)/include/embree4 -I$(GEN)/kernels -I$(GEN)/include

SCENEGRAPH := xml_parser.cpp xml_loader.cpp xml_writer.cpp obj_loader.cpp ply_loader.cpp corona_loader.cpp texture.cpp scenegraph.cpp geometry_creation.cpp
IMAGE      := image.cpp pfm.cpp ppm.cpp tga.cpp stb.cpp exr.cpp
LIGHTS     := light.cpp ambient_light.cpp directional_light.cpp point_light.cpp quad_light.cpp spot_light.cpp
TUTORIAL   := tutorial.cpp application.cpp scene.cpp tutorial_device.cpp scene_device.cpp

LIB_SRC := $(addprefix tutorials/common/scenegraph/,$(SCENEGRAPH)) $(addprefix tutorials/common/image/,$(IMAGE)) $(addprefix tutorials/common/lights/,$(LIGHTS)) \
           tutorials/common/alloc/alloc.cpp
LIB_OBJ := $(patsubst %.cpp,$(OBJ)/%.o,$(LIB_SRC))
TUT_OBJ := $(patsubst %.cpp,$(OBJ)/tutorials/common/tutorial/%.o,$(TUTORIAL))
COMMON_OBJ := $(wildcard oracle/_ref/obj/common/*/*.o)
LINK := -L$(LIBDIR) -lembree4_mi355 -Wl,-rpath,'$$ORIGIN/../../../embree_amd/lib' -Wl,--unresolved-symbols=ignore-all -lpthread -ldl

all: $(OUT)/ref_verify $(OUT)/ref_triangle_geometry

$(OBJ)/%.o: $(REF)/%.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -c $< -o $@

# (the one file of this repository in these binaries: see its header)
$(OBJ)/ref_tests_shim.o: tests/golden/ref_tests_shim.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -I$(REF) -c $< -o $@

# (the shim goes LAST: static initialisers run in link order, and its one call needs the scheduler's own statics constructed)
$(OUT)/ref_verify: $(OBJ)/tutorials/verify/verify.o $(OBJ)/tutorials/common/tutorial/application.o $(LIB_OBJ) $(OBJ)/ref_tests_shim.o
	$(CXX) -o $@ $(filter-out %ref_tests_shim.o,$^) $(COMMON_OBJ) $(OBJ)/ref_tests_shim.o $(LINK)

$(OUT)/ref_triangle_geometry: $(OBJ)/tutorials/triangle_geometry/triangle_geometry.o $(OBJ)/tutorials/triangle_geometry/triangle_geometry_device.o $(TUT_OBJ) $(LIB_OBJ) $(OBJ)/ref_tests_shim.o
	$(CXX) -o $@ $(filter-out %ref_tests_shim.o,$^) $(COMMON_OBJ) $(OBJ)/ref_tests_shim.o $(LINK)
