NAME          BOEING2
ROWS
 G  REVENUES
 G  ACOCOSTS
 N  OBJECTIV
 L  FUELAVAL
 G  SYSTDEPT
 G  ACMILES
 G  ASMILES
 G  PASSNGRS
 G  RPMILES
 G  LFRPMASM
 G  ATONMILE
 G  RTONMILE
 G  LFTNMILE
 E  FLAV*1
 E  FLAV*2
 E  FLAV*3
 E  FLAV*4
 G  LF1003S1
 G  LF1003B1
 G  LF1003C1
 G  LF1005S1
 G  LF1005B1
 G  LF1005C1
 G  LF1011S1
 G  LF1011S2
 G  LF1011B1
 G  LF1011B2
 G  LF1011C1
 G  LF1011C2
 G  LF1013S1
 G  LF1013S2
 G  LF1013B1
 G  LF1013B2
 G  LF1013C1
 G  LF1013C2
 G  LF1015S1
 G  LF1015B1
 G  LF1015C1
 G  LF1017S1
 G  LF1017B1
 G  LF1017C1
 G  LF1019S1
 G  LF1019S2
 G  LF1019B1
 G  LF1019B2
 G  LF1019C1
 G  LF1019C2
 G  LF1021S1
 G  LF1021B1
 G  LF1021C1
 G  LF1002S1
 G  LF1002B1
 G  LF1002C1
 G  LF1004S1
 G  LF1004B1
 G  LF1004C1
 G  LF1006S1
 G  LF1006S2
 G  LF1006B1
 G  LF1006B2
 G  LF1006C1
 G  LF1006C2
 G  LF1008S1
 G  LF1008S2
 G  LF1008B1
 G  LF1008B2
 G  LF1008C1
 G  LF1008C2
 G  LF1010S1
 G  LF1010S2
 G  LF1010S3
 G  LF1010B1
 G  LF1010B2
 G  LF1010B3
 G  LF1010C1
 G  LF1010C2
 G  LF1010C3
 G  LF1012S1
 G  LF1012B1
 G  LF1012C1
 G  LF1014S1
 G  LF1014S2
 G  LF1014B1
 G  LF1014B2
 G  LF1014C1
 G  LF1014C2
 G  LF1100S1
 G  LF1100S2
 G  LF1100S3
 G  LF1100S4
 G  LF1100S5
 G  LF1100B1
 G  LF1100B2
 G  LF1100B3
 G  LF1100B4
 G  LF1100B5
 G  LF1100C1
 G  LF1100C2
 G  LF1100C3
 G  LF1100C4
 G  LF1100C5
 G  LF1102S1
 G  LF1102S2
 G  LF1102S3
 G  LF1102S4
 G  LF1102B1
 G  LF1102B2
 G  LF1102B3
 G  LF1102B4
 G  LF1102C1
 G  LF1102C2
 G  LF1102C3
 G  LF1102C4
 G  LF1200S1
 G  LF1200B1
 G  LF1200C1
 G  LF1201S1
 G  LF1201B1
 G  LF1201C1
 G  NOPTCLE0
 G  CONTBOS1
 G  CONTBOS2
 G  CONTBOS3
 G  CONTBOS4
 G  CONTORD1
 G  CONTORD2
 G  CONTORD3
 G  CONTORD4
 G  CONTLGA2
 G  CONTLGA4
 G  CONTCLE1
 G  CONTCLE2
 G  CONTCLE3
 G  CONTCLE4
 L  DMBOSORD
 L  DMBOSLGA
 L  DMBOSCLE
 L  DMORDBOS
 L  DMORDLGA
 L  DMORDCLE
 L  DMLGABOS
 L  DMLGAORD
 L  DMLGACLE
 L  DMCLEBOS
 L  DMCLEORD
 L  DMCLELGA
 G  MSBOSORD
 G  MSBOSLGA
 G  MSBOSCLE
 G  MSORDBOS
 G  MSORDLGA
 G  MSORDCLE
 G  MSLGABOS
 G  MSLGAORD
 G  MSLGACLE
 G  MSCLEBOS
 G  MSCLEORD
 G  MSCLELGA
 L  DCBOSORD
 L  DCBOSCLE
 L  DCORDBOS
 L  DCORDLGA
 L  DCLGAORD
 L  DCLGACLE
 L  DCCLELGA
 G  MCORDBOS
 G  MCLGAORD
COLUMNS
    PBOSORD0  REVENUES          .075   OBJECTIV         -.075
    PBOSORD0  PASSNGRS            1.   RPMILES         .86441
    PBOSORD0  LFRPMASM       -.86441   DMBOSORD            1.
    PBOSORD0  LF1003S1           -1.
    PBOSORD1  REVENUES          .075   OBJECTIV         -.075
    PBOSORD1  PASSNGRS            1.   RPMILES         .87605
    PBOSORD1  LFRPMASM       -.87605   DMBOSORD            1.
    PBOSORD1  LF1011S1           -1.   LF1011S2           -1.
    PBOSORD2  REVENUES          .075   OBJECTIV         -.075
    PBOSORD2  PASSNGRS            1.   RPMILES         .91637
    PBOSORD2  LFRPMASM       -.91637   DMBOSORD            1.
    PBOSORD2  LF1013S1           -1.   LF1013S2           -1.
    PBOSORD3  REVENUES          .075   OBJECTIV         -.075
    PBOSORD3  PASSNGRS            1.   RPMILES         .91722
    PBOSORD3  LFRPMASM       -.91722   DMBOSORD            1.
    PBOSORD3  LF1100S1           -1.   LF1100S2           -1.
    PBOSORD3  LF1100S3           -1.
    PBOSORD4  REVENUES          .075   OBJECTIV         -.075
    PBOSORD4  PASSNGRS            1.   RPMILES         .87605
    PBOSORD4  LFRPMASM       -.87605   DMBOSORD            1.
    PBOSORD4  LF1102S1           -1.   LF1102S2           -1.
    PBOSLGA0  REVENUES          .027   OBJECTIV         -.027
    PBOSLGA0  PASSNGRS            1.   RPMILES         .18557
    PBOSLGA0  LFRPMASM       -.18557   DMBOSLGA            1.
    PBOSLGA0  LF1013S1           -1.
    PBOSLGA1  REVENUES          .027   OBJECTIV         -.027
    PBOSLGA1  PASSNGRS            1.   RPMILES         .18557
    PBOSLGA1  LFRPMASM       -.18557   DMBOSLGA            1.
    PBOSLGA1  LF1100S1           -1.
    PBOSLGA2  REVENUES          .027   OBJECTIV         -.027
    PBOSLGA2  PASSNGRS            1.   RPMILES        1.60685
    PBOSLGA2  LFRPMASM      -1.60685   DMBOSLGA            1.
    PBOSLGA2  LF1102S1           -1.   LF1102S2           -1.
    PBOSLGA2  LF1102S3           -1.
    PBOSLGA3  REVENUES          .027   OBJECTIV         -.027
    PBOSLGA3  PASSNGRS            1.   RPMILES         .18557
    PBOSLGA3  LFRPMASM       -.18557   DMBOSLGA            1.
    PBOSLGA3  LF1200S1           -1.
    PBOSCLE0  REVENUES          .053   OBJECTIV         -.053
    PBOSCLE0  PASSNGRS            1.   RPMILES         .56156
    PBOSCLE0  LFRPMASM       -.56156   DMBOSCLE            1.
    PBOSCLE0  LF1005S1           -1.
    PBOSCLE1  REVENUES          .053   OBJECTIV         -.053
    PBOSCLE1  PASSNGRS            1.   RPMILES         .56156
    PBOSCLE1  LFRPMASM       -.56156   DMBOSCLE            1.
    PBOSCLE1  LF1011S1           -1.
    PBOSCLE2  REVENUES          .053   OBJECTIV         -.053
    PBOSCLE2  PASSNGRS            1.   RPMILES         .60273
    PBOSCLE2  LFRPMASM       -.60273   DMBOSCLE            1.
    PBOSCLE2  LF1100S1           -1.   LF1100S2           -1.
    PBOSCLE3  REVENUES          .053   OBJECTIV         -.053
    PBOSCLE3  PASSNGRS            1.   RPMILES         .56156
    PBOSCLE3  LFRPMASM       -.56156   DMBOSCLE            1.
    PBOSCLE3  LF1102S1           -1.
    PORDBOS0  REVENUES          .075   OBJECTIV         -.075
    PORDBOS0  PASSNGRS            1.   RPMILES         .87605
    PORDBOS0  LFRPMASM       -.87605   DMORDBOS            1.
    PORDBOS0  LF1006S1           -1.   LF1006S2           -1.
    PORDBOS1  REVENUES          .075   OBJECTIV         -.075
    PORDBOS1  PASSNGRS            1.   RPMILES         .91637
    PORDBOS1  LFRPMASM       -.91637   DMORDBOS            1.
    PORDBOS1  LF1008S1           -1.   LF1008S2           -1.
    PORDBOS2  REVENUES          .075   OBJECTIV         -.075
    PORDBOS2  PASSNGRS            1.   RPMILES         .91722
    PORDBOS2  LFRPMASM       -.91722   DMORDBOS            1.
    PORDBOS2  LF1010S1           -1.   LF1010S2           -1.
    PORDBOS2  LF1010S3           -1.
    PORDBOS3  REVENUES          .075   OBJECTIV         -.075
    PORDBOS3  PASSNGRS            1.   RPMILES         .87605
    PORDBOS3  LFRPMASM       -.87605   DMORDBOS            1.
    PORDBOS3  LF1100S4           -1.   LF1100S5           -1.
    PORDBOS4  REVENUES          .075   OBJECTIV         -.075
    PORDBOS4  PASSNGRS            1.   RPMILES         .91637
    PORDBOS4  LFRPMASM       -.91637   DMORDBOS            1.
    PORDBOS4  LF1102S3           -1.   LF1102S4           -1.
    PORDLGA0  REVENUES          .068   OBJECTIV         -.068
    PORDLGA0  PASSNGRS            1.   RPMILES          .7308
    PORDLGA0  LFRPMASM        -.7308   DMORDLGA            1.
    PORDLGA0  LF1004S1           -1.
    PORDLGA1  REVENUES          .068   OBJECTIV         -.068
    PORDLGA1  PASSNGRS            1.   RPMILES          .7308
    PORDLGA1  LFRPMASM        -.7308   DMORDLGA            1.
    PORDLGA1  LF1008S1           -1.
    PORDLGA2  REVENUES          .068   OBJECTIV         -.068
    PORDLGA2  PASSNGRS            1.   RPMILES         .73165
    PORDLGA2  LFRPMASM       -.73165   DMORDLGA            1.
    PORDLGA2  LF1010S1           -1.   LF1010S2           -1.
    PORDLGA3  REVENUES          .068   OBJECTIV         -.068
    PORDLGA3  PASSNGRS            1.   RPMILES          .7308
    PORDLGA3  LFRPMASM        -.7308   DMORDLGA            1.
    PORDLGA3  LF1102S3           -1.
    PORDCLE0  REVENUES          .035   OBJECTIV         -.035
    PORDCLE0  PASSNGRS            1.   RPMILES          .3145
    PORDCLE0  LFRPMASM        -.3145   DMORDCLE            1.
    PORDCLE0  LF1002S1           -1.
    PORDCLE1  REVENUES          .035   OBJECTIV         -.035
    PORDCLE1  PASSNGRS            1.   RPMILES          .3145
    PORDCLE1  LFRPMASM        -.3145   DMORDCLE            1.
    PORDCLE1  LF1006S1           -1.
    PORDCLE2  REVENUES          .035   OBJECTIV         -.035
    PORDCLE2  PASSNGRS            1.   RPMILES          .3145
    PORDCLE2  LFRPMASM        -.3145   DMORDCLE            1.
    PORDCLE2  LF1010S1           -1.
    PORDCLE3  REVENUES          .035   OBJECTIV         -.035
    PORDCLE3  PASSNGRS            1.   RPMILES          .3145
    PORDCLE3  LFRPMASM        -.3145   DMORDCLE            1.
    PORDCLE3  LF1100S4           -1.
    PLGABOS0  REVENUES          .027   OBJECTIV         -.027
    PLGABOS0  PASSNGRS            1.   RPMILES         .18557
    PLGABOS0  LFRPMASM       -.18557   DMLGABOS            1.
    PLGABOS0  LF1008S2           -1.
    PLGABOS1  REVENUES          .027   OBJECTIV         -.027
    PLGABOS1  PASSNGRS            1.   RPMILES         .18557
    PLGABOS1  LFRPMASM       -.18557   DMLGABOS            1.
    PLGABOS1  LF1010S3           -1.
    PLGABOS2  REVENUES          .027   OBJECTIV         -.027
    PLGABOS2  PASSNGRS            1.   RPMILES         .18557
    PLGABOS2  LFRPMASM       -.18557   DMLGABOS            1.
    PLGABOS2  LF1014S2           -1.
    PLGABOS3  REVENUES          .027   OBJECTIV         -.027
    PLGABOS3  PASSNGRS            1.   RPMILES         1.6077
    PLGABOS3  LFRPMASM       -1.6077   DMLGABOS            1.
    PLGABOS3  LF1100S2           -1.   LF1100S3           -1.
    PLGABOS3  LF1100S4           -1.   LF1100S5           -1.
    PLGABOS4  REVENUES          .027   OBJECTIV         -.027
    PLGABOS4  PASSNGRS            1.   RPMILES         .18557
    PLGABOS4  LFRPMASM       -.18557   DMLGABOS            1.
    PLGABOS4  LF1102S4           -1.
    PLGABOS5  REVENUES          .027   OBJECTIV         -.027
    PLGABOS5  PASSNGRS            1.   RPMILES         .18557
    PLGABOS5  LFRPMASM       -.18557   DMLGABOS            1.
    PLGABOS5  LF1201S1           -1.
    PLGAORD0  REVENUES          .068   OBJECTIV         -.068
    PLGAORD0  PASSNGRS            1.   RPMILES          .7308
    PLGAORD0  LFRPMASM        -.7308   DMLGAORD            1.
    PLGAORD0  LF1013S2           -1.
    PLGAORD1  REVENUES          .068   OBJECTIV         -.068
    PLGAORD1  PASSNGRS            1.   RPMILES          .7308
    PLGAORD1  LFRPMASM        -.7308   DMLGAORD            1.
    PLGAORD1  LF1017S1           -1.
    PLGAORD2  REVENUES          .068   OBJECTIV         -.068
    PLGAORD2  PASSNGRS            1.   RPMILES         .73165
    PLGAORD2  LFRPMASM       -.73165   DMLGAORD            1.
    PLGAORD2  LF1019S1           -1.   LF1019S2           -1.
    PLGAORD3  REVENUES          .068   OBJECTIV         -.068
    PLGAORD3  PASSNGRS            1.   RPMILES         .73165
    PLGAORD3  LFRPMASM       -.73165   DMLGAORD            1.
    PLGAORD3  LF1100S2           -1.   LF1100S3           -1.
    PLGACLE0  REVENUES          .037   OBJECTIV         -.037
    PLGACLE0  PASSNGRS            1.   RPMILES         .41715
    PLGACLE0  LFRPMASM       -.41715   DMLGACLE            1.
    PLGACLE0  LF1015S1           -1.
    PLGACLE1  REVENUES          .037   OBJECTIV         -.037
    PLGACLE1  PASSNGRS            1.   RPMILES         .41715
    PLGACLE1  LFRPMASM       -.41715   DMLGACLE            1.
    PLGACLE1  LF1019S1           -1.
    PLGACLE2  REVENUES          .037   OBJECTIV         -.037
    PLGACLE2  PASSNGRS            1.   RPMILES         .41715
    PLGACLE2  LFRPMASM       -.41715   DMLGACLE            1.
    PLGACLE2  LF1100S2           -1.
    PCLEBOS0  REVENUES          .053   OBJECTIV         -.053
    PCLEBOS0  PASSNGRS            1.   RPMILES         .56156
    PCLEBOS0  LFRPMASM       -.56156   DMCLEBOS            1.
    PCLEBOS0  LF1006S2           -1.
    PCLEBOS1  REVENUES          .053   OBJECTIV         -.053
    PCLEBOS1  PASSNGRS            1.   RPMILES         .60273
    PCLEBOS1  LFRPMASM       -.60273   DMCLEBOS            1.
    PCLEBOS1  LF1010S2           -1.   LF1010S3           -1.
    PCLEBOS2  REVENUES          .053   OBJECTIV         -.053
    PCLEBOS2  PASSNGRS            1.   RPMILES         .60273
    PCLEBOS2  LFRPMASM       -.60273   DMCLEBOS            1.
    PCLEBOS2  LF1014S1           -1.   LF1014S2           -1.
    PCLEBOS3  REVENUES          .053   OBJECTIV         -.053
    PCLEBOS3  PASSNGRS            1.   RPMILES         .56156
    PCLEBOS3  LFRPMASM       -.56156   DMCLEBOS            1.
    PCLEBOS3  LF1100S5           -1.
    PCLEBOS4  REVENUES          .053   OBJECTIV         -.053
    PCLEBOS4  PASSNGRS            1.   RPMILES        1.23087
    PCLEBOS4  LFRPMASM      -1.23087   DMCLEBOS            1.
    PCLEBOS4  LF1102S2           -1.   LF1102S3           -1.
    PCLEBOS4  LF1102S4           -1.
    PCLEORD0  REVENUES          .035   OBJECTIV         -.035
    PCLEORD0  PASSNGRS            1.   RPMILES          .3145
    PCLEORD0  LFRPMASM        -.3145   DMCLEORD            1.
    PCLEORD0  LF1011S2           -1.
    PCLEORD1  REVENUES          .035   OBJECTIV         -.035
    PCLEORD1  PASSNGRS            1.   RPMILES          .3145
    PCLEORD1  LFRPMASM        -.3145   DMCLEORD            1.
    PCLEORD1  LF1019S2           -1.
    PCLEORD2  REVENUES          .035   OBJECTIV         -.035
    PCLEORD2  PASSNGRS            1.   RPMILES          .3145
    PCLEORD2  LFRPMASM        -.3145   DMCLEORD            1.
    PCLEORD2  LF1021S1           -1.
    PCLEORD3  REVENUES          .035   OBJECTIV         -.035
    PCLEORD3  PASSNGRS            1.   RPMILES          .3145
    PCLEORD3  LFRPMASM        -.3145   DMCLEORD            1.
    PCLEORD3  LF1100S3           -1.
    PCLEORD4  REVENUES          .035   OBJECTIV         -.035
    PCLEORD4  PASSNGRS            1.   RPMILES          .3145
    PCLEORD4  LFRPMASM        -.3145   DMCLEORD            1.
    PCLEORD4  LF1102S2           -1.
    PCLELGA0  REVENUES          .037   OBJECTIV         -.037
    PCLELGA0  PASSNGRS            1.   RPMILES         .41715
    PCLELGA0  LFRPMASM       -.41715   DMCLELGA            1.
    PCLELGA0  LF1010S2           -1.
    PCLELGA1  REVENUES          .037   OBJECTIV         -.037
    PCLELGA1  PASSNGRS            1.   RPMILES         .41715
    PCLELGA1  LFRPMASM       -.41715   DMCLELGA            1.
    PCLELGA1  LF1012S1           -1.
    PCLELGA2  REVENUES          .037   OBJECTIV         -.037
    PCLELGA2  PASSNGRS            1.   RPMILES         .41715
    PCLELGA2  LFRPMASM       -.41715   DMCLELGA            1.
    PCLELGA2  LF1014S1           -1.
    PCLELGA3  REVENUES          .037   OBJECTIV         -.037
    PCLELGA3  PASSNGRS            1.   RPMILES         1.0453
    PCLELGA3  LFRPMASM       -1.0453   DMCLELGA            1.
    PCLELGA3  LF1102S2           -1.   LF1102S3           -1.
    BBOSORD0  REVENUES           .75   OBJECTIV          -.75
    BBOSORD0  RTONMILE        .86441   LFTNMILE       -.86441
    BBOSORD0  DCBOSORD            1.   LF1003B1           -1.
    BBOSORD1  REVENUES           .75   OBJECTIV          -.75
    BBOSORD1  RTONMILE        .87605   LFTNMILE       -.87605
    BBOSORD1  DCBOSORD            1.   LF1011B1           -1.
    BBOSORD1  LF1011B2           -1.
    BBOSCLE0  REVENUES           .53   OBJECTIV          -.53
    BBOSCLE0  RTONMILE        .56156   LFTNMILE       -.56156
    BBOSCLE0  DCBOSCLE            1.   LF1005B1           -1.
    BBOSCLE1  REVENUES           .53   OBJECTIV          -.53
    BBOSCLE1  RTONMILE        .56156   LFTNMILE       -.56156
    BBOSCLE1  DCBOSCLE            1.   LF1011B1           -1.
    BORDBOS0  REVENUES           .75   OBJECTIV          -.75
    BORDBOS0  RTONMILE        .87605   LFTNMILE       -.87605
    BORDBOS0  DCORDBOS            1.   LF1006B1           -1.
    BORDBOS0  LF1006B2           -1.
    CBOSORD0  REVENUES           .75   OBJECTIV          -.75
    CBOSORD0  RTONMILE        .86441   LFTNMILE       -.86441
    CBOSORD0  DCBOSORD            1.   LF1003C1           -1.
    CBOSORD1  REVENUES           .75   OBJECTIV          -.75
    CBOSORD1  RTONMILE        .87605   LFTNMILE       -.87605
    CBOSORD1  DCBOSORD            1.   LF1011C1           -1.
    CBOSORD1  LF1011C2           -1.
    CBOSORD2  REVENUES           .75   OBJECTIV          -.75
    CBOSORD2  RTONMILE        .91637   LFTNMILE       -.91637
    CBOSORD2  DCBOSORD            1.   LF1013C1           -1.
    CBOSORD2  LF1013C2           -1.
    CBOSORD3  REVENUES           .75   OBJECTIV          -.75
    CBOSORD3  RTONMILE        .91722   LFTNMILE       -.91722
    CBOSORD3  DCBOSORD            1.   LF1100C1           -1.
    CBOSORD3  LF1100C2           -1.   LF1100C3           -1.
    CBOSORD4  REVENUES           .75   OBJECTIV          -.75
    CBOSORD4  RTONMILE        .87605   LFTNMILE       -.87605
    CBOSORD4  DCBOSORD            1.   LF1102C1           -1.
    CBOSORD4  LF1102C2           -1.
    CBOSCLE0  REVENUES           .53   OBJECTIV          -.53
    CBOSCLE0  RTONMILE        .56156   LFTNMILE       -.56156
    CBOSCLE0  DCBOSCLE            1.   LF1005C1           -1.
    CBOSCLE1  REVENUES           .53   OBJECTIV          -.53
    CBOSCLE1  RTONMILE        .56156   LFTNMILE       -.56156
    CBOSCLE1  DCBOSCLE            1.   LF1011C1           -1.
    CBOSCLE2  REVENUES           .53   OBJECTIV          -.53
    CBOSCLE2  RTONMILE        .60273   LFTNMILE       -.60273
    CBOSCLE2  DCBOSCLE            1.   LF1100C1           -1.
    CBOSCLE2  LF1100C2           -1.
    CBOSCLE3  REVENUES           .53   OBJECTIV          -.53
    CBOSCLE3  RTONMILE        .56156   LFTNMILE       -.56156
    CBOSCLE3  DCBOSCLE            1.   LF1102C1           -1.
    CORDBOS0  REVENUES           .75   OBJECTIV          -.75
    CORDBOS0  RTONMILE        .87605   LFTNMILE       -.87605
    CORDBOS0  DCORDBOS            1.   LF1006C1           -1.
    CORDBOS0  LF1006C2           -1.
    CORDBOS1  REVENUES           .75   OBJECTIV          -.75
    CORDBOS1  RTONMILE        .91637   LFTNMILE       -.91637
    CORDBOS1  DCORDBOS            1.   LF1008C1           -1.
    CORDBOS1  LF1008C2           -1.
    CORDBOS2  REVENUES           .75   OBJECTIV          -.75
    CORDBOS2  RTONMILE        .91722   LFTNMILE       -.91722
    CORDBOS2  DCORDBOS            1.   LF1010C1           -1.
    CORDBOS2  LF1010C2           -1.   LF1010C3           -1.
    CORDBOS3  REVENUES           .75   OBJECTIV          -.75
    CORDBOS3  RTONMILE        .87605   LFTNMILE       -.87605
    CORDBOS3  DCORDBOS            1.   LF1100C4           -1.
    CORDBOS3  LF1100C5           -1.
    CORDBOS4  REVENUES           .75   OBJECTIV          -.75
    CORDBOS4  RTONMILE        .91637   LFTNMILE       -.91637
    CORDBOS4  DCORDBOS            1.   LF1102C3           -1.
    CORDBOS4  LF1102C4           -1.
    CORDLGA0  REVENUES           .68   OBJECTIV          -.68
    CORDLGA0  RTONMILE         .7308   LFTNMILE        -.7308
    CORDLGA0  DCORDLGA            1.   LF1004C1           -1.
    CORDLGA1  REVENUES           .68   OBJECTIV          -.68
    CORDLGA1  RTONMILE         .7308   LFTNMILE        -.7308
    CORDLGA1  DCORDLGA            1.   LF1008C1           -1.
    CORDLGA2  REVENUES           .68   OBJECTIV          -.68
    CORDLGA2  RTONMILE        .73165   LFTNMILE       -.73165
    CORDLGA2  DCORDLGA            1.   LF1010C1           -1.
    CORDLGA2  LF1010C2           -1.
    CORDLGA3  REVENUES           .68   OBJECTIV          -.68
    CORDLGA3  RTONMILE         .7308   LFTNMILE        -.7308
    CORDLGA3  DCORDLGA            1.   LF1102C3           -1.
    CLGAORD0  REVENUES           .68   OBJECTIV          -.68
    CLGAORD0  RTONMILE         .7308   LFTNMILE        -.7308
    CLGAORD0  DCLGAORD            1.   LF1013C2           -1.
    CLGAORD1  REVENUES           .68   OBJECTIV          -.68
    CLGAORD1  RTONMILE         .7308   LFTNMILE        -.7308
    CLGAORD1  DCLGAORD            1.   LF1017C1           -1.
    CLGAORD2  REVENUES           .68   OBJECTIV          -.68
    CLGAORD2  RTONMILE        .73165   LFTNMILE       -.73165
    CLGAORD2  DCLGAORD            1.   LF1019C1           -1.
    CLGAORD2  LF1019C2           -1.
    CLGAORD3  REVENUES           .68   OBJECTIV          -.68
    CLGAORD3  RTONMILE        .73165   LFTNMILE       -.73165
    CLGAORD3  DCLGAORD            1.   LF1100C2           -1.
    CLGAORD3  LF1100C3           -1.
    CLGACLE0  REVENUES           .37   OBJECTIV          -.37
    CLGACLE0  RTONMILE        .41715   LFTNMILE       -.41715
    CLGACLE0  DCLGACLE            1.   LF1015C1           -1.
    CLGACLE1  REVENUES           .37   OBJECTIV          -.37
    CLGACLE1  RTONMILE        .41715   LFTNMILE       -.41715
    CLGACLE1  DCLGACLE            1.   LF1019C1           -1.
    CLGACLE2  REVENUES           .37   OBJECTIV          -.37
    CLGACLE2  RTONMILE        .41715   LFTNMILE       -.41715
    CLGACLE2  DCLGACLE            1.   LF1100C2           -1.
    CCLELGA0  REVENUES           .37   OBJECTIV          -.37
    CCLELGA0  RTONMILE        .41715   LFTNMILE       -.41715
    CCLELGA0  DCCLELGA            1.   LF1010C2           -1.
    CCLELGA1  REVENUES           .37   OBJECTIV          -.37
    CCLELGA1  RTONMILE        .41715   LFTNMILE       -.41715
    CCLELGA1  DCCLELGA            1.   LF1012C1           -1.
    CCLELGA2  REVENUES           .37   OBJECTIV          -.37
    CCLELGA2  RTONMILE        .41715   LFTNMILE       -.41715
    CCLELGA2  DCCLELGA            1.   LF1014C1           -1.
    CCLELGA3  REVENUES           .37   OBJECTIV          -.37
    CCLELGA3  RTONMILE        1.0453   LFTNMILE       -1.0453
    CCLELGA3  DCCLELGA            1.   LF1102C2           -1.
    CCLELGA3  LF1102C3           -1.
    GRDTIMO1  FLAV*1              1.   ACOCOSTS           .65
    GRDTIMO1  OBJECTIV           .65
    GRDTIMN1  FLAV*1              1.   ACOCOSTS          -.05
    GRDTIMN1  OBJECTIV          -.05
    GRDTIMO2  FLAV*2              1.   ACOCOSTS          .275
    GRDTIMO2  OBJECTIV          .275
    GRDTIMN2  FLAV*2              1.   ACOCOSTS       -.02889
    GRDTIMN2  OBJECTIV       -.02889
    GRDTIMN3  FLAV*3              1.   ACOCOSTS       -.03611
    GRDTIMN3  OBJECTIV       -.03611
    GRDTIMN4  FLAV*4              1.   ACOCOSTS       -.01333
    GRDTIMN4  OBJECTIV       -.01333
    N1003AC1  ACOCOSTS           .01   OBJECTIV           .01
    N1003AC1  FUELAVAL       7.98429   SYSTDEPT            1.
    N1003AC1  ACMILES         .86441   ASMILES      351.81396
    N1003AC1  LFRPMASM    211.088376   FLAV*1         2.32729
    N1003AC1  ATONMILE      25.93224   LFTNMILE      12.96612
    N1003AC1  LF1003S1          305.   LF1003B1           12.
    N1003AC1  CONTBOS1            1.   CONTORD1           -1.
    N1003AC1  MSBOSORD            1.
    N1003AC2  ACOCOSTS           .02   OBJECTIV           .02
    N1003AC2  FUELAVAL       2.51914   SYSTDEPT            1.
    N1003AC2  ACMILES         .86441   ASMILES      113.23743
    N1003AC2  LFRPMASM     67.942458   FLAV*2         2.10966
    N1003AC2  LF1003S1           98.   CONTBOS2            1.
    N1003AC2  CONTORD2           -1.   MSBOSORD            1.
    N1003AC3  ACOCOSTS           .03   OBJECTIV           .03
    N1003AC3  FUELAVAL       3.15178   SYSTDEPT            1.
    N1003AC3  ACMILES         .86441   FLAV*3         2.10966
    N1003AC3  ATONMILE      34.57631   LFTNMILE     17.288155
    N1003AC3  LF1003C1           30.   CONTBOS3            1.
    N1003AC3  CONTORD3           -1.
    N1003AC4  ACOCOSTS           .04   OBJECTIV           .04
    N1003AC4  FUELAVAL       1.99337   SYSTDEPT            1.
    N1003AC4  ACMILES         .86441   FLAV*4         2.10966
    N1003AC4  ATONMILE      12.96612   LFTNMILE       6.48306
    N1003AC4  LF1003C1         11.25   CONTBOS4            1.
    N1003AC4  CONTORD4           -1.
    N1005AC1  ACOCOSTS       4.04337   OBJECTIV       4.04337
    N1005AC1  FUELAVAL       5.83404   SYSTDEPT            1.
    N1005AC1  ACMILES         .56156   ASMILES      228.55299
    N1005AC1  LFRPMASM    137.131794   FLAV*1           1000.
    N1005AC1  ATONMILE      16.84665   LFTNMILE      8.423325
    N1005AC1  LF1005S1          305.   LF1005B1           12.
    N1005AC1  NOPTCLE0            1.   CONTBOS1            1.
    N1005AC1  CONTCLE1           -1.   MSBOSCLE            1.
    N1005AC2  ACOCOSTS       1.60964   OBJECTIV       1.60964
    N1005AC2  FUELAVAL       1.82258   SYSTDEPT            1.
    N1005AC2  ACMILES         .56156   ASMILES       73.56374
    N1005AC2  LFRPMASM     44.138244   FLAV*2           2000.
    N1005AC2  LF1005S1           98.   NOPTCLE0            1.
    N1005AC2  CONTBOS2            1.   CONTCLE2           -1.
    N1005AC2  MSBOSCLE            1.
    N1005AC3  ACOCOSTS       1.93119   OBJECTIV       1.93119
    N1005AC3  FUELAVAL       2.27351   SYSTDEPT            1.
    N1005AC3  ACMILES         .56156   FLAV*3           3000.
    N1005AC3  ATONMILE       22.4622   LFTNMILE       11.2311
    N1005AC3  LF1005C1           30.   NOPTCLE0            1.
    N1005AC3  CONTBOS3            1.   CONTCLE3           -1.
    N1005AC4  ACOCOSTS       1.34618   OBJECTIV       1.34618
    N1005AC4  FUELAVAL       1.41795   SYSTDEPT            1.
    N1005AC4  ACMILES         .56156   FLAV*4           1500.
    N1005AC4  ATONMILE       8.42333   LFTNMILE      4.211665
    N1005AC4  LF1005C1         11.25   NOPTCLE0            1.
    N1005AC4  CONTBOS4            1.   CONTCLE4           -1.
    N1011AC1  ACOCOSTS           1.1   OBJECTIV           1.1
    N1011AC1  FUELAVAL       9.91398   SYSTDEPT            2.
    N1011AC1  ACMILES         .87605   ASMILES      356.55371
    N1011AC1  LFRPMASM    213.932226   FLAV*1             25.
    N1011AC1  ATONMILE       26.2816   LFTNMILE       13.1408
    N1011AC1  LF1011S1          305.   LF1011S2          305.
    N1011AC1  LF1011B1           12.   LF1011B2           12.
    N1011AC1  NOPTCLE0            2.   CONTBOS1            1.
    N1011AC1  CONTORD1           -1.   MSBOSCLE            1.
    N1011AC1  MSBOSORD            1.   MSCLEORD            1.
    N1011AC2  ACOCOSTS           1.2   OBJECTIV           1.2
    N1011AC2  FUELAVAL       3.07692   SYSTDEPT            2.
    N1011AC2  ACMILES         .87605   ASMILES      114.76299
    N1011AC2  LFRPMASM     68.857794   FLAV*2             25.
    N1011AC2  LF1011S1           98.   LF1011S2           98.
    N1011AC2  NOPTCLE0            2.   CONTBOS2            1.
    N1011AC2  CONTORD2           -1.   MSBOSCLE            1.
    N1011AC2  MSBOSORD            1.   MSCLEORD            1.
    N1011AC3  ACOCOSTS           1.3   OBJECTIV           1.3
    N1011AC3  FUELAVAL       3.83055   SYSTDEPT            2.
    N1011AC3  ACMILES         .87605   FLAV*3             26.
    N1011AC3  ATONMILE      35.04214   LFTNMILE      17.52107
    N1011AC3  LF1011C1           30.   LF1011C2           30.
    N1011AC3  NOPTCLE0            2.   CONTBOS3            1.
    N1011AC3  CONTORD3           -1.
    N1011AC4  ACOCOSTS       2.34647   OBJECTIV       2.34647
    N1011AC4  FUELAVAL        2.3665   SYSTDEPT            2.
    N1011AC4  ACMILES         .87605   FLAV*4             27.
    N1011AC4  ATONMILE       13.1408   LFTNMILE        6.5704
    N1011AC4  LF1011C1         11.25   LF1011C2         11.25
    N1011AC4  NOPTCLE0            2.   CONTBOS4            1.
    N1011AC4  CONTORD4           -1.
    N1013AC2  ACOCOSTS       2.36783   OBJECTIV       2.36783
    N1013AC2  FUELAVAL       3.16965   SYSTDEPT            2.
    N1013AC2  ACMILES         .91637   ASMILES      120.04449
    N1013AC2  LFRPMASM     72.026694   FLAV*2         2.65943
    N1013AC2  LF1013S1           98.   LF1013S2           98.
    N1013AC2  CONTBOS2            1.   CONTORD2           -1.
    N1013AC2  MSBOSLGA            1.   MSBOSORD            1.
    N1013AC2  MSLGAORD            1.
    N1013AC4  ACOCOSTS       1.90292   OBJECTIV       1.90292
    N1013AC4  FUELAVAL        2.4431   SYSTDEPT            2.
    N1013AC4  ACMILES         .91637   FLAV*4         2.65943
    N1013AC4  ATONMILE      13.74556   LFTNMILE       6.87278
    N1013AC4  LF1013C1         11.25   LF1013C2         11.25
    N1013AC4  CONTBOS4            1.   CONTORD4           -1.
    N1013AC4  MCLGAORD            1.
    N1015AC2  ACOCOSTS       1.36416   OBJECTIV       1.36416
    N1015AC2  FUELAVAL       1.49045   SYSTDEPT            1.
    N1015AC2  ACMILES         .41715   ASMILES       54.64705
    N1015AC2  LFRPMASM      32.78823   FLAV*2         1.25093
    N1015AC2  LF1015S1           98.   NOPTCLE0            1.
    N1015AC2  CONTLGA2            1.   CONTCLE2           -1.
    N1015AC2  MSLGACLE            1.
    N1015AC4  ACOCOSTS       1.14401   OBJECTIV       1.14401
    N1015AC4  FUELAVAL       1.14359   SYSTDEPT            1.
    N1015AC4  ACMILES         .41715   FLAV*4         1.25093
    N1015AC4  ATONMILE       6.25729   LFTNMILE      3.128645
    N1015AC4  LF1015C1         11.25   NOPTCLE0            1.
    N1015AC4  CONTLGA4            1.   CONTCLE4           -1.
    N1017AC2  ACOCOSTS       1.64736   OBJECTIV       1.64736
    N1017AC2  FUELAVAL       2.21183   SYSTDEPT            1.
    N1017AC2  ACMILES          .7308   ASMILES       95.73444
    N1017AC2  LFRPMASM     57.440664   FLAV*2         1.85313
    N1017AC2  LF1017S1           98.   CONTLGA2            1.
    N1017AC2  CONTORD2           -1.   MSLGAORD            1.
    N1017AC4  ACOCOSTS       1.33312   OBJECTIV       1.33312
    N1017AC4  FUELAVAL       1.73951   SYSTDEPT            1.
    N1017AC4  ACMILES          .7308   FLAV*4         1.85313
    N1017AC4  ATONMILE      10.96196   LFTNMILE       5.48098
    N1017AC4  LF1017C1         11.25   CONTLGA4            1.
    N1017AC4  CONTORD4           -1.   MCLGAORD            1.
    N1019AC2  ACOCOSTS       2.55381   OBJECTIV       2.55381
    N1019AC2  FUELAVAL        2.7448   SYSTDEPT            2.
    N1019AC2  ACMILES         .73165   ASMILES        95.8463
    N1019AC2  LFRPMASM      57.50778   FLAV*2         2.30477
    N1019AC2  LF1019S1           98.   LF1019S2           98.
    N1019AC2  NOPTCLE0            2.   CONTLGA2            1.
    N1019AC2  CONTORD2           -1.   MSLGACLE            1.
    N1019AC2  MSLGAORD            1.   MSCLEORD            1.
    N1019AC4  ACOCOSTS       2.14431   OBJECTIV       2.14431
    N1019AC4  FUELAVAL       2.09214   SYSTDEPT            2.
    N1019AC4  ACMILES         .73165   FLAV*4         2.30477
    N1019AC4  ATONMILE      10.97477   LFTNMILE      5.487385
    N1019AC4  LF1019C1         11.25   LF1019C2         11.25
    N1019AC4  NOPTCLE0            2.   CONTLGA4            1.
    N1019AC4  CONTORD4           -1.   MCLGAORD            1.
    N1021AC1  ACOCOSTS       3.12679   OBJECTIV       3.12679
    N1021AC1  FUELAVAL       4.07994   SYSTDEPT            1.
    N1021AC1  ACMILES          .3145   ASMILES      128.00075
    N1021AC1  LFRPMASM      76.80045   FLAV*1         1.34295
    N1021AC1  ATONMILE       9.43495   LFTNMILE      4.717475
    N1021AC1  LF1021S1          305.   LF1021B1           12.
    N1021AC1  NOPTCLE0            1.   CONTCLE1            1.
    N1021AC1  CONTORD1           -1.   MSCLEORD            1.
    N1021AC2  ACOCOSTS       1.18965   OBJECTIV       1.18965
    N1021AC2  FUELAVAL       1.25435   SYSTDEPT            1.
    N1021AC2  ACMILES          .3145   ASMILES       41.19926
    N1021AC2  LFRPMASM     24.719556   FLAV*2         1.05384
    N1021AC2  LF1021S1           98.   NOPTCLE0            1.
    N1021AC2  CONTCLE2            1.   CONTORD2           -1.
    N1021AC2  MSCLEORD            1.
    N1021AC3  ACOCOSTS       1.42472   OBJECTIV       1.42472
    N1021AC3  FUELAVAL       1.55704   SYSTDEPT            1.
    N1021AC3  ACMILES          .3145   FLAV*3         1.05384
    N1021AC3  ATONMILE      12.57993   LFTNMILE      6.289965
    N1021AC3  LF1021C1           30.   NOPTCLE0            1.
    N1021AC3  CONTCLE3            1.   CONTORD3           -1.
    N1021AC4  ACOCOSTS        1.0003   OBJECTIV        1.0003
    N1021AC4  FUELAVAL        .94855   SYSTDEPT            1.
    N1021AC4  ACMILES          .3145   FLAV*4         1.05384
    N1021AC4  ATONMILE       4.71747   LFTNMILE      2.358735
    N1021AC4  LF1021C1         11.25   NOPTCLE0            1.
    N1021AC4  CONTCLE4            1.   CONTORD4           -1.
    N1002AC1  ACOCOSTS       3.12679   OBJECTIV       3.12679
    N1002AC1  FUELAVAL       4.07994   SYSTDEPT            1.
    N1002AC1  ACMILES          .3145   ASMILES      128.00075
    N1002AC1  LFRPMASM      76.80045   FLAV*1         1.34295
    N1002AC1  ATONMILE       9.43495   LFTNMILE      4.717475
    N1002AC1  LF1002S1          305.   LF1002B1           12.
    N1002AC1  NOPTCLE0            1.   CONTORD1            1.
    N1002AC1  CONTCLE1           -1.   MSORDCLE            1.
    N1002AC2  ACOCOSTS       1.18965   OBJECTIV       1.18965
    N1002AC2  FUELAVAL       1.25435   SYSTDEPT            1.
    N1002AC2  ACMILES          .3145   ASMILES       41.19926
    N1002AC2  LFRPMASM     24.719556   FLAV*2         1.05384
    N1002AC2  LF1002S1           98.   NOPTCLE0            1.
    N1002AC2  CONTORD2            1.   CONTCLE2           -1.
    N1002AC2  MSORDCLE            1.
    N1002AC3  ACOCOSTS       1.42472   OBJECTIV       1.42472
    N1002AC3  FUELAVAL       1.55704   SYSTDEPT            1.
    N1002AC3  ACMILES          .3145   FLAV*3         1.05384
    N1002AC3  ATONMILE      12.57993   LFTNMILE      6.289965
    N1002AC3  LF1002C1           30.   NOPTCLE0            1.
    N1002AC3  CONTORD3            1.   CONTCLE3           -1.
    N1002AC4  ACOCOSTS        1.0003   OBJECTIV        1.0003
    N1002AC4  FUELAVAL        .94855   SYSTDEPT            1.
    N1002AC4  ACMILES          .3145   FLAV*4         1.05384
    N1002AC4  ATONMILE       4.71747   LFTNMILE      2.358735
    N1002AC4  LF1002C1         11.25   NOPTCLE0            1.
    N1002AC4  CONTORD4            1.   CONTCLE4           -1.
    N1004AC2  ACOCOSTS       1.64736   OBJECTIV       1.64736
    N1004AC2  FUELAVAL       2.21183   SYSTDEPT            1.
    N1004AC2  ACMILES          .7308   ASMILES       95.73444
    N1004AC2  LFRPMASM     57.440664   FLAV*2         1.85313
    N1004AC2  LF1004S1           98.   CONTORD2            1.
    N1004AC2  CONTLGA2           -1.   MSORDLGA            1.
    N1004AC4  ACOCOSTS       1.33312   OBJECTIV       1.33312
    N1004AC4  FUELAVAL       1.73951   SYSTDEPT            1.
    N1004AC4  ACMILES          .7308   FLAV*4         1.85313
    N1004AC4  ATONMILE      10.96196   LFTNMILE       5.48098
    N1004AC4  LF1004C1         11.25   CONTORD4            1.
    N1004AC4  CONTLGA4           -1.
    N1006AC1  ACOCOSTS       7.17016   OBJECTIV       7.17016
    N1006AC1  FUELAVAL       9.91398   SYSTDEPT            2.
    N1006AC1  ACMILES         .87605   ASMILES      356.55371
    N1006AC1  LFRPMASM    213.932226   FLAV*1         3.12813
    N1006AC1  ATONMILE       26.2816   LFTNMILE       13.1408
    N1006AC1  LF1006S1          305.   LF1006S2          305.
    N1006AC1  LF1006B1           12.   LF1006B2           12.
    N1006AC1  NOPTCLE0            2.   CONTORD1            1.
    N1006AC1  CONTBOS1           -1.   MSORDCLE            1.
    N1006AC1  MSORDBOS            1.   MSCLEBOS            1.
    N1006AC2  ACOCOSTS       2.79929   OBJECTIV       2.79929
    N1006AC2  FUELAVAL       3.07692   SYSTDEPT            2.
    N1006AC2  ACMILES         .87605   ASMILES      114.76299
    N1006AC2  LFRPMASM     68.857794   FLAV*2         2.58202
    N1006AC2  LF1006S1           98.   LF1006S2           98.
    N1006AC2  NOPTCLE0            2.   CONTORD2            1.
    N1006AC2  CONTBOS2           -1.   MSORDCLE            1.
    N1006AC2  MSORDBOS            1.   MSCLEBOS            1.
    N1006AC3  ACOCOSTS       3.35591   OBJECTIV       3.35591
    N1006AC3  FUELAVAL       3.83055   SYSTDEPT            2.
    N1006AC3  ACMILES         .87605   FLAV*3         2.58202
    N1006AC3  ATONMILE      35.04214   LFTNMILE      17.52107
    N1006AC3  LF1006C1           30.   LF1006C2           30.
    N1006AC3  NOPTCLE0            2.   CONTORD3            1.
    N1006AC3  CONTBOS3           -1.   MCORDBOS            1.
    N1006AC4  ACOCOSTS       2.34647   OBJECTIV       2.34647
    N1006AC4  FUELAVAL        2.3665   SYSTDEPT            2.
    N1006AC4  ACMILES         .87605   FLAV*4         2.58202
    N1006AC4  ATONMILE       13.1408   LFTNMILE        6.5704
    N1006AC4  LF1006C1         11.25   LF1006C2         11.25
    N1006AC4  NOPTCLE0            2.   CONTORD4            1.
    N1006AC4  CONTBOS4           -1.   MCORDBOS            1.
    N1008AC2  ACOCOSTS       2.36783   OBJECTIV       2.36783
    N1008AC2  FUELAVAL       3.16965   SYSTDEPT            2.
    N1008AC2  ACMILES         .91637   ASMILES      120.04449
    N1008AC2  LFRPMASM     72.026694   FLAV*2         2.65943
    N1008AC2  LF1008S1           98.   LF1008S2           98.
    N1008AC2  CONTORD2            1.   CONTBOS2           -1.
    N1008AC2  MSORDLGA            1.   MSORDBOS            1.
    N1008AC2  MSLGABOS            1.
    N1008AC4  ACOCOSTS       1.90292   OBJECTIV       1.90292
    N1008AC4  FUELAVAL        2.4431   SYSTDEPT            2.
    N1008AC4  ACMILES         .91637   FLAV*4         2.65943
    N1008AC4  ATONMILE      13.74556   LFTNMILE       6.87278
    N1008AC4  LF1008C1         11.25   LF1008C2         11.25
    N1008AC4  CONTORD4            1.   CONTBOS4           -1.
    N1008AC4  MCORDBOS            1.
    N1010AC2  ACOCOSTS       3.27428   OBJECTIV       3.27428
    N1010AC2  FUELAVAL       3.70262   SYSTDEPT            3.
    N1010AC2  ACMILES         .91722   ASMILES      120.15637
    N1010AC2  LFRPMASM     72.093822   FLAV*2         3.11107
    N1010AC2  LF1010S1           98.   LF1010S2           98.
    N1010AC2  LF1010S3           98.   NOPTCLE0            2.
    N1010AC2  CONTORD2            1.   CONTBOS2           -1.
    N1010AC2  MSORDCLE            1.   MSORDLGA            1.
    N1010AC2  MSORDBOS            1.   MSCLELGA            1.
    N1010AC2  MSCLEBOS            1.   MSLGABOS            1.
    N1010AC4  ACOCOSTS       2.71411   OBJECTIV       2.71411
    N1010AC4  FUELAVAL       2.79573   SYSTDEPT            3.
    N1010AC4  ACMILES         .91722   FLAV*4         3.11107
    N1010AC4  ATONMILE      13.75836   LFTNMILE       6.87918
    N1010AC4  LF1010C1         11.25   LF1010C2         11.25
    N1010AC4  LF1010C3         11.25   NOPTCLE0            2.
    N1010AC4  CONTORD4            1.   CONTBOS4           -1.
    N1010AC4  MCORDBOS            1.
    N1012AC2  ACOCOSTS       1.36416   OBJECTIV       1.36416
    N1012AC2  FUELAVAL       1.49045   SYSTDEPT            1.
    N1012AC2  ACMILES         .41715   ASMILES       54.64705
    N1012AC2  LFRPMASM      32.78823   FLAV*2         1.25093
    N1012AC2  LF1012S1           98.   NOPTCLE0            1.
    N1012AC2  CONTCLE2            1.   CONTLGA2           -1.
    N1012AC2  MSCLELGA            1.
    N1012AC4  ACOCOSTS       1.14401   OBJECTIV       1.14401
    N1012AC4  FUELAVAL       1.14359   SYSTDEPT            1.
    N1012AC4  ACMILES         .41715   FLAV*4         1.25093
    N1012AC4  ATONMILE       6.25729   LFTNMILE      3.128645
    N1012AC4  LF1012C1         11.25   NOPTCLE0            1.
    N1012AC4  CONTCLE4            1.   CONTLGA4           -1.
    N1014AC2  ACOCOSTS       2.08463   OBJECTIV       2.08463
    N1014AC2  FUELAVAL       2.44827   SYSTDEPT            2.
    N1014AC2  ACMILES         .60273   ASMILES       78.95706
    N1014AC2  LFRPMASM     47.374236   FLAV*2         2.05723
    N1014AC2  LF1014S1           98.   LF1014S2           98.
    N1014AC2  NOPTCLE0            1.   CONTCLE2            1.
    N1014AC2  CONTBOS2           -1.   MSCLELGA            1.
    N1014AC2  MSCLEBOS            1.   MSLGABOS            1.
    N1014AC4  ACOCOSTS       1.71382   OBJECTIV       1.71382
    N1014AC4  FUELAVAL       1.84718   SYSTDEPT            2.
    N1014AC4  ACMILES         .60273   FLAV*4         2.05723
    N1014AC4  ATONMILE       9.04089   LFTNMILE      4.520445
    N1014AC4  LF1014C1         11.25   LF1014C2         11.25
    N1014AC4  NOPTCLE0            1.   CONTCLE4            1.
    N1014AC4  CONTBOS4           -1.
    N1100AC2  ACOCOSTS       6.07357   OBJECTIV       6.07357
    N1100AC2  FUELAVAL       6.77953   SYSTDEPT            5.
    N1100AC2  ACMILES        1.79328   ASMILES      234.91937
    N1100AC2  LFRPMASM    140.951622   FLAV*2         5.69309
    N1100AC2  LF1100S1           98.   LF1100S2           98.
    N1100AC2  LF1100S3           98.   LF1100S4           98.
    N1100AC2  LF1100S5           98.   NOPTCLE0            4.
    N1100AC2  MSBOSLGA            1.   MSBOSCLE            1.
    N1100AC2  MSBOSORD            1.   MSLGACLE            1.
    N1100AC2  MSLGAORD            1.   MSLGABOS            1.
    N1100AC2  MSCLEORD            1.   MSORDCLE            1.
    N1100AC2  MSORDBOS            1.   MSCLEBOS            1.
    N1100AC4  ACOCOSTS       5.06059   OBJECTIV       5.06059
    N1100AC4  FUELAVAL       5.16223   SYSTDEPT            5.
    N1100AC4  ACMILES        1.79328   FLAV*4         5.69309
    N1100AC4  ATONMILE      26.89915   LFTNMILE     13.449575
    N1100AC4  LF1100C1         11.25   LF1100C2         11.25
    N1100AC4  LF1100C3         11.25   LF1100C4         11.25
    N1100AC4  LF1100C5         11.25   NOPTCLE0            4.
    N1100AC4  MCLGAORD            1.   MCORDBOS            1.
    N1102AC2  ACOCOSTS       5.16712   OBJECTIV       5.16712
    N1102AC2  FUELAVAL       6.24657   SYSTDEPT            4.
    N1102AC2  ACMILES        1.79242   ASMILES      234.80756
    N1102AC2  LFRPMASM    140.884536   FLAV*2         5.24145
    N1102AC2  LF1102S1           98.   LF1102S2           98.
    N1102AC2  LF1102S3           98.   LF1102S4           98.
    N1102AC2  NOPTCLE0            2.   MSBOSCLE            1.
    N1102AC2  MSBOSORD            1.   MSBOSLGA            1.
    N1102AC2  MSCLEORD            1.   MSCLELGA            1.
    N1102AC2  MSCLEBOS            1.   MSORDLGA            1.
    N1102AC2  MSORDBOS            1.   MSLGABOS            1.
    N1102AC4  ACOCOSTS       4.24939   OBJECTIV       4.24939
    N1102AC4  FUELAVAL        4.8096   SYSTDEPT            4.
    N1102AC4  ACMILES        1.79242   FLAV*4         5.24145
    N1102AC4  ATONMILE      26.88635   LFTNMILE     13.443175
    N1102AC4  LF1102C1         11.25   LF1102C2         11.25
    N1102AC4  LF1102C3         11.25   LF1102C4         11.25
    N1102AC4  NOPTCLE0            2.   MCORDBOS            1.
    N1200AC2  ACOCOSTS        .72047   OBJECTIV        .72047
    N1200AC2  FUELAVAL        .95782   SYSTDEPT            1.
    N1200AC2  ACMILES         .18557   ASMILES       24.31007
    N1200AC2  LFRPMASM     14.586042   FLAV*2           .8063
    N1200AC2  LF1200S1           98.   CONTBOS2            1.
    N1200AC2  CONTLGA2           -1.   MSBOSLGA            1.
    N1200AC4  ACOCOSTS         .5698   OBJECTIV         .5698
    N1200AC4  FUELAVAL        .70359   SYSTDEPT            1.
    N1200AC4  ACMILES         .18557   FLAV*4           .8063
    N1200AC4  ATONMILE        2.7836   LFTNMILE        1.3918
    N1200AC4  LF1200C1         11.25   CONTBOS4            1.
    N1200AC4  CONTLGA4           -1.
    N1201AC2  ACOCOSTS        .72047   OBJECTIV        .72047
    N1201AC2  FUELAVAL        .95782   SYSTDEPT            1.
    N1201AC2  ACMILES         .18557   ASMILES       24.31007
    N1201AC2  LFRPMASM     14.586042   FLAV*2           .8063
    N1201AC2  LF1201S1           98.   CONTLGA2            1.
    N1201AC2  CONTBOS2           -1.   MSLGABOS            1.
    N1201AC4  ACOCOSTS         .5698   OBJECTIV         .5698
    N1201AC4  FUELAVAL        .70359   SYSTDEPT            1.
    N1201AC4  ACMILES         .18557   FLAV*4           .8063
    N1201AC4  ATONMILE        2.7836   LFTNMILE        1.3918
    N1201AC4  LF1201C1         11.25   CONTLGA4            1.
    N1201AC4  CONTBOS4           -1.
RHS
    RHS1      FUELAVAL       100000.   PASSNGRS         9431.
    RHS1      SYSTDEPT           50.   FLAV*1             30.
    RHS1      FLAV*2             45.   DMBOSORD          302.
    RHS1      DMBOSLGA         2352.   DMBOSCLE          142.
    RHS1      DMORDBOS          302.   DMORDLGA          515.
    RHS1      DMORDCLE          619.   DMLGABOS         2743.
    RHS1      DMLGAORD          712.   DMLGACLE          517.
    RHS1      DMCLEBOS          131.   DMCLEORD          712.
    RHS1      DMCLELGA          409.   MSBOSORD            3.
    RHS1      MSBOSLGA            7.   MSBOSCLE            1.
    RHS1      MSORDBOS            3.   MSORDLGA            4.
    RHS1      MSORDCLE            5.   MSLGABOS            7.
    RHS1      MSLGAORD            5.   MSLGACLE            4.
    RHS1      MSCLEBOS            1.   MSCLEORD            6.
    RHS1      MSCLELGA            3.   MCORDBOS            1.
    RHS1      MCLGAORD            2.   DCBOSORD           12.
    RHS1      DCBOSCLE           16.   DCORDBOS           24.
    RHS1      DCORDLGA           13.   DCLGAORD           45.
    RHS1      DCLGACLE           16.   DCCLELGA            5.
    RHS1      NOPTCLE0           24.
RANGES
    RANGE1    DMBOSORD           61.   DMBOSLGA          471.
    RANGE1    DMBOSCLE           29.   DMORDBOS           61.
    RANGE1    DMORDLGA          103.   DMORDCLE          124.
    RANGE1    DMLGABOS          549.   DMLGAORD          143.
    RANGE1    DMLGACLE          104.   DMCLEBOS           27.
    RANGE1    DMCLEORD          143.   DMCLELGA           82.
    RANGE1    DCBOSORD           12.   DCBOSCLE           3.2
    RANGE1    DCORDBOS           4.8   DCORDLGA           2.6
    RANGE1    DCLGAORD            9.   DCLGACLE           3.2
    RANGE1    DCCLELGA            5.
BOUNDS
 LO INTBOU    GRDTIMN1         -100.
 UP INTBOU    GRDTIMN1            0.
 LO INTBOU    GRDTIMN2          -90.
 UP INTBOU    GRDTIMN2            0.
 LO INTBOU    GRDTIMN3          -45.
 UP INTBOU    GRDTIMN3            0.
 LO INTBOU    GRDTIMN4          -45.
 UP INTBOU    GRDTIMN4            0.
 UP INTBOU    N1003AC1            7.
 UP INTBOU    N1003AC2            7.
 UP INTBOU    N1003AC3            2.
 UP INTBOU    N1003AC4            2.
 UP INTBOU    N1005AC1            7.
 UP INTBOU    N1005AC2            7.
 UP INTBOU    N1005AC3            2.
 UP INTBOU    N1005AC4            2.
 UP INTBOU    N1011AC1            7.
 UP INTBOU    N1011AC2            7.
 UP INTBOU    N1011AC3            2.
 UP INTBOU    N1011AC4            2.
 UP INTBOU    N1013AC2           14.
 UP INTBOU    N1013AC4            2.
 UP INTBOU    N1015AC2            7.
 UP INTBOU    N1015AC4            2.
 UP INTBOU    N1017AC2            7.
 UP INTBOU    N1017AC4            2.
 UP INTBOU    N1019AC2            7.
 UP INTBOU    N1019AC4            2.
 UP INTBOU    N1021AC1            7.
 UP INTBOU    N1021AC2            7.
 UP INTBOU    N1021AC3            2.
 UP INTBOU    N1021AC4            2.
 UP INTBOU    N1002AC1            7.
 UP INTBOU    N1002AC2            7.
 UP INTBOU    N1002AC3            2.
 UP INTBOU    N1002AC4            2.
 UP INTBOU    N1004AC2            7.
 UP INTBOU    N1004AC4            2.
 UP INTBOU    N1006AC1            7.
 UP INTBOU    N1006AC2            7.
 UP INTBOU    N1006AC3            2.
 UP INTBOU    N1006AC4            2.
 UP INTBOU    N1008AC2           14.
 UP INTBOU    N1008AC4            2.
 UP INTBOU    N1010AC2           14.
 UP INTBOU    N1010AC4            2.
 UP INTBOU    N1012AC2            7.
 UP INTBOU    N1012AC4            2.
 UP INTBOU    N1014AC2           14.
 UP INTBOU    N1014AC4            2.
 UP INTBOU    N1100AC2            7.
 UP INTBOU    N1100AC4            7.
 UP INTBOU    N1102AC2            7.
 UP INTBOU    N1102AC4            7.
 UP INTBOU    N1200AC2           14.
 UP INTBOU    N1200AC4            7.
 UP INTBOU    N1201AC2           14.
 UP INTBOU    N1201AC4            7.
ENDATA
