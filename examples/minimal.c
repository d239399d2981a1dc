/* The smallest application of the Embree 4 C API on this library: one triangle, two rays through rtcIntersect1 (the known answers of the reference's
 * tutorials/minimal: a hit at t = 1 on geometry 0 / primitive 0, and a miss), then the same two rays and an occlusion query through the batched
 * extension rtcIntersect1M / rtcOccluded1M.  Plain C, nothing but include/embree4/rtcore.h:
 *     gcc -std=c11 -Iinclude examples/minimal.c -Lembree_amd/lib -lembree4_mi355 -Wl,-rpath,$PWD/embree_amd/lib -lm -o minimal && ./minimal
 * An application written against Embree itself builds the same way. */
#include <embree4/rtcore.h>
#include <math.h>
#include <stdio.h>

static void report(void* userPtr, enum RTCError code, const char* str) { (void)userPtr; printf("error %d: %s\n", (int)code, str); }

static void set_ray(struct RTCRay* r, float ox, float oy, float oz, float dx, float dy, float dz) {
  r->org_x = ox; r->org_y = oy; r->org_z = oz; r->dir_x = dx; r->dir_y = dy; r->dir_z = dz;
  r->tnear = 0.0f; r->tfar = INFINITY; r->time = 0.0f; r->mask = 0xFFFFFFFFu; r->id = 0; r->flags = 0;
}

int main(void) {
  RTCDevice device = rtcNewDevice(NULL);
  if (!device) { printf("no device: error %d (this library needs a HIP device, it has no CPU fallback)\n", (int)rtcGetDeviceError(NULL)); return 1; }
  rtcSetDeviceErrorFunction(device, report, NULL);

  RTCScene scene = rtcNewScene(device);
  RTCGeometry geom = rtcNewGeometry(device, RTC_GEOMETRY_TYPE_TRIANGLE);
  float* v = (float*)rtcSetNewGeometryBuffer(geom, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, 3 * sizeof(float), 3);
  unsigned* t = (unsigned*)rtcSetNewGeometryBuffer(geom, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, 3 * sizeof(unsigned), 1);
  if (!v || !t) return 2;
  v[0] = 0; v[1] = 0; v[2] = 0;  v[3] = 1; v[4] = 0; v[5] = 0;  v[6] = 0; v[7] = 1; v[8] = 0;
  t[0] = 0; t[1] = 1; t[2] = 2;
  rtcCommitGeometry(geom);
  unsigned geomID = rtcAttachGeometry(scene, geom);
  rtcReleaseGeometry(geom);
  rtcCommitScene(scene);

  struct RTCRayHit rh[2];
  set_ray(&rh[0].ray, 0.33f, 0.33f, -1.0f, 0, 0, 1);     /* through the triangle */
  set_ray(&rh[1].ray, 1.00f, 1.00f, -1.0f, 0, 0, 1);     /* past it */
  for (int i = 0; i < 2; i++) { rh[i].hit.geomID = RTC_INVALID_GEOMETRY_ID; rh[i].hit.instID[0] = RTC_INVALID_GEOMETRY_ID; }
  for (int i = 0; i < 2; i++) {
    rtcIntersect1(scene, &rh[i], NULL);
    if (rh[i].hit.geomID != RTC_INVALID_GEOMETRY_ID) printf("ray %d: hit geomID %u primID %u tfar %g u %g v %g\n", i, rh[i].hit.geomID, rh[i].hit.primID, rh[i].ray.tfar, rh[i].hit.u, rh[i].hit.v);
    else printf("ray %d: no hit\n", i);
  }

  /* the batched extension: M records in one call (what a renderer's wavefront loop uses) */
  struct RTCRayHit batch[2];
  set_ray(&batch[0].ray, 0.33f, 0.33f, -1.0f, 0, 0, 1); set_ray(&batch[1].ray, 1.00f, 1.00f, -1.0f, 0, 0, 1);
  for (int i = 0; i < 2; i++) { batch[i].hit.geomID = RTC_INVALID_GEOMETRY_ID; batch[i].hit.instID[0] = RTC_INVALID_GEOMETRY_ID; }
  rtcIntersect1M(scene, batch, 2, sizeof(struct RTCRayHit), NULL);
  printf("batch: %s %s\n", batch[0].hit.geomID == geomID && batch[0].ray.tfar == 1.0f ? "hit" : "?", batch[1].hit.geomID == RTC_INVALID_GEOMETRY_ID ? "miss" : "?");
  struct RTCRay shadow[2];
  set_ray(&shadow[0], 0.33f, 0.33f, -1.0f, 0, 0, 1); set_ray(&shadow[1], 1.00f, 1.00f, -1.0f, 0, 0, 1);
  rtcOccluded1M(scene, shadow, 2, sizeof(struct RTCRay), NULL);
  printf("occluded: %s %s\n", shadow[0].tfar < 0.0f ? "yes" : "no", shadow[1].tfar < 0.0f ? "yes" : "no");

  struct RTCBounds b; rtcGetSceneBounds(scene, &b);
  printf("bounds: %g %g %g .. %g %g %g\n", b.lower_x, b.lower_y, b.lower_z, b.upper_x, b.upper_y, b.upper_z);
  rtcReleaseScene(scene);
  rtcReleaseDevice(device);
  printf("errors: %d\n", (int)rtcGetDeviceError(NULL));
  return 0;
}
