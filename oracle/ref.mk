# oracle/ref.mk -- TEST INFRASTRUCTURE (never linked or loaded by the product path).
#
# Builds the *real* reference (Embree 4.4.1) straight from the sources where
# they lie under $(REF) with plain g++, as a single-ISA (AVX2) shared library:
#     oracle/_ref/libembree4.so
# No file of the reference is copied into this repository; all outputs
# (objects, the three instantiated config headers, the .so) go to oracle/_ref/
# which is git-ignored.  The reference's own build system (CMake) is NOT run:
# this file re-states the source lists of kernels/CMakeLists.txt:35-118 (main
# library) and :120-211 (macro embree_files, evaluated for ISA == ISA_LOWEST ==
# AVX2) and the flags of common/cmake/gnu.cmake:19,41-70,91-92.
#
# The three headers CMake would CONFIGURE_FILE (CMakeLists.txt:659-682) are
# instantiated from the reference's own .in templates with sed, default options
# (all geometry types on => default ABI, sizeof(RTCRayHit)==96).
#
#   make -f oracle/ref.mk -j8            # ~3-4 min on 8 cores
REF   ?= /root/reference
ROOT_OUT ?= oracle/_ref
# ISA=avx2 (default): oracle/_ref/libembree4.so + libref_driver.so.   ISA=avx512: a second single-ISA library, oracle/_ref/avx512/libembree4_avx512.so +
# libref_driver_avx512.so (other file names: two libraries of one name cannot live in one process), flags of common/cmake/gnu.cmake:20 (FLAGS_AVX512) and
# the two 16-wide packet files kernels/CMakeLists.txt:198-202 adds above AVX2.  BASELINE.md asks for both on the CPU side of the comparison.
ISA   ?= avx2
ifeq ($(ISA),avx512)
OUT   ?= $(ROOT_OUT)/avx512
SUFFIX := _avx512
ISAFLAGS := -march=skylake-avx512
ISADEF := -DEMBREE_TARGET_AVX512
else
OUT   ?= $(ROOT_OUT)
SUFFIX :=
ISAFLAGS := -mf16c -mavx2 -mfma -mlzcnt -mbmi -mbmi2
ISADEF := -DEMBREE_TARGET_AVX2
endif
GEN   := $(OUT)/gen
CXX   ?= g++

CXXFLAGS := -std=c++11 -O3 -DNDEBUG -fPIC -fsigned-char -flax-vector-conversions \
            -fno-strict-overflow -fno-delete-null-pointer-checks -fwrapv \
            -fvisibility=hidden -fvisibility-inlines-hidden -fno-strict-aliasing \
            -fno-tree-vectorize -w $(ISAFLAGS) \
            -DTASKING_INTERNAL $(ISADEF) \
            -I$(GEN)/kernels/common -I$(GEN)/kernels/bvh -I$(GEN)/include/embree4 \
            -I$(GEN)/kernels -I$(GEN)/include

COMMON_SRC := \
  common/sys/sysinfo.cpp common/sys/alloc.cpp common/sys/filename.cpp common/sys/library.cpp \
  common/sys/thread.cpp common/sys/estring.cpp common/sys/regression.cpp common/sys/mutex.cpp \
  common/sys/condition.cpp common/sys/barrier.cpp \
  common/math/constants.cpp common/simd/sse.cpp \
  common/lexers/stringstream.cpp common/lexers/tokenstream.cpp \
  common/tasking/taskschedulerinternal.cpp

MAIN_SRC := \
  common/device.cpp common/stat.cpp common/acceln.cpp common/accelset.cpp common/state.cpp \
  common/rtcore.cpp common/rtcore_builder.cpp common/scene.cpp common/scene_verify.cpp \
  common/alloc.cpp common/geometry.cpp common/scene_user_geometry.cpp common/scene_instance.cpp \
  common/scene_instance_array.cpp common/scene_triangle_mesh.cpp common/scene_quad_mesh.cpp \
  common/scene_curves.cpp common/scene_line_segments.cpp common/scene_grid_mesh.cpp \
  common/scene_points.cpp common/motion_derivative.cpp \
  subdiv/bezier_curve.cpp subdiv/bspline_curve.cpp subdiv/catmullrom_curve.cpp \
  geometry/primitive4.cpp geometry/instance_intersector.cpp geometry/instance_array_intersector.cpp \
  geometry/curve_intersector_virtual_4v.cpp geometry/curve_intersector_virtual_4i.cpp \
  geometry/curve_intersector_virtual_4i_mb.cpp geometry/curve_intersector_virtual_8v.cpp \
  geometry/curve_intersector_virtual_8i.cpp geometry/curve_intersector_virtual_8i_mb.cpp \
  builders/primrefgen.cpp \
  bvh/bvh.cpp bvh/bvh_statistics.cpp bvh/bvh4_factory.cpp bvh/bvh8_factory.cpp \
  bvh/bvh_collider.cpp bvh/bvh_rotate.cpp bvh/bvh_refit.cpp bvh/bvh_builder.cpp \
  bvh/bvh_builder_hair.cpp bvh/bvh_builder_hair_mb.cpp bvh/bvh_builder_morton.cpp \
  bvh/bvh_builder_sah.cpp bvh/bvh_builder_sah_spatial.cpp bvh/bvh_builder_sah_mb.cpp \
  bvh/bvh_builder_twolevel.cpp bvh/bvh_intersector1_bvh4.cpp \
  common/scene_subdiv_mesh.cpp subdiv/tessellation_cache.cpp subdiv/subdivpatch1base.cpp \
  subdiv/catmullclark_coefficients.cpp geometry/grid_soa.cpp subdiv/subdivpatch1base_eval.cpp \
  bvh/bvh_builder_subdiv.cpp \
  bvh/bvh_intersector_hybrid4_bvh4.cpp

# kernels/CMakeLists.txt:120-211 for ISA==AVX2==ISA_LOWEST, minus the files
# that the main library already provides in namespace `isa` when the lowest ISA
# is AVX2 (those would only be re-compiled into the same namespace).
ISA_SRC := \
  geometry/primitive8.cpp \
  bvh/bvh_intersector1_bvh8.cpp \
  bvh/bvh_intersector_hybrid8_bvh4.cpp bvh/bvh_intersector_hybrid4_bvh8.cpp \
  bvh/bvh_intersector_hybrid8_bvh8.cpp
ifeq ($(ISA),avx512)
ISA_SRC += bvh/bvh_intersector_hybrid16_bvh8.cpp bvh/bvh_intersector_hybrid16_bvh4.cpp
endif

COMMON_OBJ := $(patsubst %.cpp,$(OUT)/obj/%.o,$(COMMON_SRC))
MAIN_OBJ   := $(patsubst %.cpp,$(OUT)/obj/kernels/%.o,$(MAIN_SRC))
ISA_OBJ    := $(patsubst %.cpp,$(OUT)/obj/kernels_avx2/%.o,$(ISA_SRC))

all: $(OUT)/libembree4$(SUFFIX).so $(OUT)/libref_driver$(SUFFIX).so

HDRS := $(GEN)/kernels/config.h $(GEN)/kernels/hash.h $(GEN)/include/embree4/rtcore_config.h

$(GEN)/.dirs:
	mkdir -p $(GEN)/kernels/common $(GEN)/kernels/bvh $(GEN)/include/embree4
	touch $@

# "#cmakedefine X" -> "#define X" for every option that defaults ON
# (CMakeLists.txt:185-214), "/* #undef X */" otherwise.
ON := EMBREE_RAY_MASK EMBREE_FILTER_FUNCTION EMBREE_GEOMETRY_TRIANGLE EMBREE_GEOMETRY_QUAD \
      EMBREE_GEOMETRY_CURVE EMBREE_GEOMETRY_SUBDIVISION EMBREE_GEOMETRY_USER EMBREE_GEOMETRY_INSTANCE \
      EMBREE_GEOMETRY_INSTANCE_ARRAY EMBREE_GEOMETRY_GRID EMBREE_GEOMETRY_POINT EMBREE_RAY_PACKETS \
      EMBREE_DISC_POINT_SELF_INTERSECTION_AVOIDANCE
SED_ON := $(foreach o,$(ON),-e 's/^\#cmakedefine $(o)$$/\#define $(o)/')

$(GEN)/kernels/config.h: $(REF)/kernels/config.h.in $(GEN)/.dirs
	sed $(SED_ON) -e 's/^#cmakedefine \(.*\)$$/\/* #undef \1 *\//' \
	    -e 's/@EMBREE_CURVE_SELF_INTERSECTION_AVOIDANCE_FACTOR@/2.0/' $< > $@

$(GEN)/include/embree4/rtcore_config.h: $(REF)/kernels/rtcore_config.h.in $(GEN)/.dirs
	sed $(SED_ON) -e 's/^#cmakedefine01 \(.*\)$$/#define \1 0/' \
	    -e 's/^#cmakedefine \(.*\)$$/\/* #undef \1 *\//' \
	    -e 's/@EMBREE_VERSION_MAJOR@/4/g' -e 's/@EMBREE_VERSION_MINOR@/4/g' \
	    -e 's/@EMBREE_VERSION_PATCH@/1/g' -e 's/@EMBREE_VERSION_NUMBER@/40401/g' \
	    -e 's/@EMBREE_VERSION_NOTE@//g' -e 's/@EMBREE_MAX_INSTANCE_LEVEL_COUNT@/1/g' \
	    -e 's/@EMBREE_API_NAMESPACE@//g' $< > $@

$(GEN)/kernels/hash.h: $(REF)/kernels/hash.h.in $(GEN)/.dirs
	sed -e 's/@EMBREE_HASH@/oracle-ref-build/' $< > $@

$(OUT)/obj/common/%.o: $(REF)/common/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -DEMBREE_LOWEST_ISA -c $< -o $@

$(OUT)/obj/kernels/%.o: $(REF)/kernels/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -DEMBREE_LOWEST_ISA -c $< -o $@

$(OUT)/obj/kernels_avx2/%.o: $(REF)/kernels/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -c $< -o $@

# export only the rtc* C API, like kernels/export.linux.map does
$(OUT)/export.map:
	@mkdir -p $(OUT)
	printf '{ global: rtc*; local: *; };\n' > $@

$(OUT)/libembree4$(SUFFIX).so: $(COMMON_OBJ) $(MAIN_OBJ) $(ISA_OBJ) $(OUT)/export.map
	$(CXX) -shared -o $@ $(COMMON_OBJ) $(MAIN_OBJ) $(ISA_OBJ) \
	    -Wl,--version-script=$(OUT)/export.map -lpthread -ldl

clean:
	rm -rf $(OUT)
.PHONY: all clean

# C-ABI shim used by tests/bench through ctypes (see oracle/ref_driver.cpp)
driver: $(OUT)/libref_driver$(SUFFIX).so
$(OUT)/libref_driver$(SUFFIX).so: oracle/ref_driver.cpp $(OUT)/libembree4$(SUFFIX).so
	$(CXX) -std=c++17 -O2 -fPIC -shared -mavx2 -o $@ $< -I$(REF)/include -I$(GEN)/include/embree4 \
	    -I$(GEN)/include -L$(OUT) -lembree4$(SUFFIX) -Wl,-rpath,'$$ORIGIN' -lpthread
.PHONY: driver
