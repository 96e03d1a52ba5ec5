/* lizard_xxhash.h — private declarations of lizard_xxhash.c for the other host files (the frame layer); the public ones are
 * in include/lizard_amd.h.  State layouts stay within the reference's XXH32_state_t / XXH64_state_t (48 / 88 bytes,
 * lib/xxhash/xxhash.h:257-279): callers allocate those on their own stack. */
#ifndef LIZARD_XXHASH_H
#define LIZARD_XXHASH_H
#include <stddef.h>
#include <stdint.h>

typedef struct Lizard_XXH32_state_s { uint32_t total32, large, v[4]; uint8_t buf[16]; uint32_t fill, reserved; } Lizard_XXH32_state_t;   /* 48 bytes */
typedef struct Lizard_XXH64_state_s { uint64_t total, v[4]; uint8_t buf[32]; uint32_t fill, reserved; } Lizard_XXH64_state_t;           /* 80 bytes */

#ifdef __cplusplus
extern "C" {
#endif
unsigned Lizard_XXH_versionNumber(void);
unsigned Lizard_XXH32(const void* input, size_t length, unsigned seed);
Lizard_XXH32_state_t* Lizard_XXH32_createState(void);
int      Lizard_XXH32_freeState(Lizard_XXH32_state_t* statePtr);
void     Lizard_XXH32_copyState(Lizard_XXH32_state_t* dst, const Lizard_XXH32_state_t* src);
int      Lizard_XXH32_reset(Lizard_XXH32_state_t* statePtr, unsigned seed);
int      Lizard_XXH32_update(Lizard_XXH32_state_t* statePtr, const void* input, size_t length);
unsigned Lizard_XXH32_digest(const Lizard_XXH32_state_t* statePtr);
void     Lizard_XXH32_canonicalFromHash(unsigned char* dst, unsigned hash);
unsigned Lizard_XXH32_hashFromCanonical(const unsigned char* src);
unsigned long long Lizard_XXH64(const void* input, size_t length, unsigned long long seed);
Lizard_XXH64_state_t* Lizard_XXH64_createState(void);
int      Lizard_XXH64_freeState(Lizard_XXH64_state_t* statePtr);
void     Lizard_XXH64_copyState(Lizard_XXH64_state_t* dst, const Lizard_XXH64_state_t* src);
int      Lizard_XXH64_reset(Lizard_XXH64_state_t* statePtr, unsigned long long seed);
int      Lizard_XXH64_update(Lizard_XXH64_state_t* statePtr, const void* input, size_t length);
unsigned long long Lizard_XXH64_digest(const Lizard_XXH64_state_t* statePtr);
void     Lizard_XXH64_canonicalFromHash(unsigned char* dst, unsigned long long hash);
unsigned long long Lizard_XXH64_hashFromCanonical(const unsigned char* src);
#ifdef __cplusplus
}
#endif
#endif
